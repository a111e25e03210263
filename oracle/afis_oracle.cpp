// afis_oracle.cpp — CPU restatement of the MSU-LatentAFIS matcher hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (msu-latentafis_amd/, the
// `match` CLI, libafis_hip.so) may include, link or call this file.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// PARITY STATUS
//   * S4 (per-query PQ look-up table) and the T1 data model (texture code extraction) are
//     PINNED: tests compare this file against oracle/_ref/libafis_ref.so, which is compiled from
//     the reference's own header /root/reference/matching/include.h (self-contained, no
//     third-party dependency).
//   * The PQ encoder (orc_pq_encode; SURVEY §8f-1, extraction/descriptor_PQ.py:19-27) is PINNED to scipy.cluster.vq.vq — the
//     routine the reference calls — through tests/golden/golden_pq.npz (made by tests/golden/make_golden_pq.py).
//   * Everything that lives in /root/reference/matching/matcher.cpp (S1-S3, S5-S11, F1, F2) is
//     "PARITY UNPINNED": matcher.cpp needs Eigen and Boost.Filesystem, which are neither in the
//     reference tree nor in this image, the reference ships no tests or golden vectors for this
//     path, and the three scores in sample_data/sample_scores.txt need the Python-2/TF-1.3
//     extraction stack to reproduce.  Those stages are restated line by line below with the
//     reference's evaluation order, float/double promotions and thresholds, each citing the
//     file:line it follows.
//
// Where the reference's arithmetic order is not its own (Eigen GEMM / mat-vec / reductions,
// matcher.cpp:443,455-456,1286-1288,1408-1410 — Eigen version unpinned) this file fixes a
// canonical order: k-ascending fmaf chain for the descriptor GEMM, index-ascending unfused
// mul+add for mat-vecs and sums.  The HIP path uses the same canonical orders.
//
// Every scorer's `tie_mode` argument carries two fields: bits 0-3 the tie mode below, bits 4.. the accumulation
// order of the Eigen-owned sums (0 = canonical; see "accumulation orders" further down).
// tie_mode: 0 = std::sort on the reference's comparator (what the reference does; the order of
//               equal keys is whatever libstdc++'s introsort yields),
//           1 = canonical: equal keys ordered by ascending index (std::stable_sort).
//               This is the order the HIP path implements.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <string>
#include <tuple>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

#define ORC_PI 3.1415926  // include.h:22 (double literal; every comparison against it is in double)

struct Point { int x, y; float ori; };  // include.h:24-31 (reliability unused on this path)

struct MinuTpl {          // include.h:203-252
    int n = 0, des_len = 0;
    std::vector<Point> pts;
    std::vector<float> des;  // [n][des_len]
};
struct LatTexTpl {        // include.h:298-364
    int n = 0, des_len = 0;
    std::vector<Point> pts;
    std::vector<float> des;  // [n][des_len]
    std::vector<float> lut;  // [n][M][K]   (m_dist_codewords)
};
struct RolTexTpl {        // include.h:366-485
    int n = 0, des_len = 0;
    std::vector<Point> pts;
    std::vector<uint8_t> codes;  // [n][des_len]
};
struct Latent { std::vector<MinuTpl> minu; std::vector<LatTexTpl> tex; int load_rc = 0; };
struct Rolled { std::vector<MinuTpl> minu; std::vector<RolTexTpl> tex; int load_rc = 0; };

struct Codebook {         // matcher.cpp:31-94
    int M = 0, K = 0, dsub = 0;
    std::vector<float> cw;          // [M][K][dsub]
    std::vector<float> table_dist;  // [50*50]
    int dist_N = 50;
    int N = 200;
};

typedef std::tuple<float, int, int> Corr;  // (similarity, latent idx, rolled idx)

// ---- sorting helper -------------------------------------------------------------------------
// site: which of the reference's sorts this is — 1 = S3 (:476), 2 = S7 (:741), 4 = S8's selection (:1301, :1423), 8 = S9's (:1590).  Tie modes 2..5 take the
// reference's std::sort at SOME sites and the stable order at the others (2: S9; 3: S8 + S9; 4: S3; 5: S7; 6: S3 + S7; 7: S3 + S8; 8: S3 + S9; 9: S3 + S8 + S9 = what the HIP path delivers with option ref_tie_order 2): they exist to measure which site's tie order moves scores
// (tools/tie_site_sweep.py).  Mode 4 is also what the HIP path implements with option s3_tie_order 1 (csrc/stdsort_order.h): the GPU tests compare that option with it bit for bit.
enum { kSiteS3 = 1, kSiteS7 = 2, kSiteS8 = 4, kSiteS9 = 8 };
template <class Cmp>
void sort_idx(std::vector<int>& y, Cmp cmp, int tie_mode, int site)
{
    const int t = tie_mode & 15;
    const int unstable_sites = t == 0 ? 15 : t == 2 ? kSiteS9 : t == 3 ? (kSiteS8 | kSiteS9) : t == 4 ? kSiteS3 : t == 5 ? kSiteS7 : t == 6 ? (kSiteS3 | kSiteS7) : t == 7 ? (kSiteS3 | kSiteS8) : t == 8 ? (kSiteS3 | kSiteS9) : t == 9 ? (kSiteS3 | kSiteS8 | kSiteS9) : 0;
    if (unstable_sites & site) std::sort(y.begin(), y.end(), cmp);
    else std::stable_sort(y.begin(), y.end(), cmp);
}

// ---- accumulation orders (sum_order = bits 4.. of the `tie_mode` argument every scorer takes) ---------------------------------
// The reference leaves three pieces of arithmetic to Eigen (version unpinned, not in the tree): the descriptor product of S1
// (matcher.cpp:443), the row / column sums of S2 (:455-456) and the mat-vec + sum of the S8 power iteration (:1286-1288, :1408-1410).
// Order 0 is the canonical one the HIP path implements.  The others exist to MEASURE how far a score can move when the same sums are
// taken in the orders an Eigen build takes them (tools/order_sweep.py -> profiles/r06_order_sweep.json; the parity claim of DESIGN
// section 2 rests on that measurement, not on an argument):
//   0  canonical: S1 = k-ascending fmaf chain; every other sum index-ascending, multiply and add rounded separately
//   1  S1 = k-ascending, multiply and add rounded separately (the GEBP kernel of the reference's own build flags, -O3 without -march:
//      SSE2, no FMA; GEBP vectorises over OUTPUTS, each output's sum runs over k in order); the rest as 0
//   2  as 1, and the contiguous reductions (S2's row sums, S8's mat-vec rows and its sum of c) as 4 strided partial sums reduced
//      (l0 + l2) + (l1 + l3) at the end, scalar tail added last — Eigen's linear vectorised redux / row-major gemv with SSE packets;
//      S2's column sums stay sequential (strided in memory: not vectorised along the sum)
//   3  every sum, S1 included, as 4 strided partial sums (unfused)
//   4  every sum as 8 strided partial sums with fused multiply-adds (an -march=native AVX2 + FMA build), reduced 8 -> 4 -> 2 -> 1
//   5  every sum pairwise (recursive halving), unfused
static inline float lanes_reduce(const float* l, int W)
{
    float q[4];
    if (W == 8) { for (int i = 0; i < 4; ++i) q[i] = l[i] + l[i + 4]; } else { for (int i = 0; i < 4; ++i) q[i] = l[i]; }
    float a = q[0] + q[2], b = q[1] + q[3];
    return a + b;
}
static float pairwise_sum(const float* x, int n, int stride)
{
    if (n <= 0) return 0.f;
    if (n == 1) return x[0];
    if (n == 2) return x[0] + x[stride];
    int h = n / 2;
    float a = pairwise_sum(x, h, stride), b = pairwise_sum(x + (size_t)h * stride, n - h, stride);
    return a + b;
}
// sum_i x[i * stride]; how: 0 sequential, 4 / 8 strided lanes, -1 pairwise
static float sum_how(const float* x, int n, int stride, int how)
{
    if (how == -1) return pairwise_sum(x, n, stride);
    if (how == 0) { float s = 0.f; for (int i = 0; i < n; ++i) s += x[(size_t)i * stride]; return s; }
    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int W = how, full = n - n % W;
    for (int i = 0; i < full; ++i) l[i % W] += x[(size_t)i * stride];
    float s = lanes_reduce(l, W);
    for (int i = full; i < n; ++i) s += x[(size_t)i * stride];
    return s;
}
// sum_k a[k] * b[k]; fused: fmaf
static float dot_how(const float* a, const float* b, int n, int how, bool fused)
{
    if (how == -1) { std::vector<float> p(n); for (int k = 0; k < n; ++k) p[k] = a[k] * b[k]; return pairwise_sum(p.data(), n, 1); }
    if (how == 0) {
        float acc = 0.f;
        if (fused) { for (int k = 0; k < n; ++k) acc = fmaf(a[k], b[k], acc); }
        else { for (int k = 0; k < n; ++k) { float p = a[k] * b[k]; acc += p; } }
        return acc;
    }
    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int W = how, full = n - n % W;
    for (int k = 0; k < full; ++k) { if (fused) l[k % W] = fmaf(a[k], b[k], l[k % W]); else { float p = a[k] * b[k]; l[k % W] += p; } }
    float s = lanes_reduce(l, W);
    for (int k = full; k < n; ++k) { if (fused) s = fmaf(a[k], b[k], s); else { float p = a[k] * b[k]; s += p; } }
    return s;
}
struct SumOrder {
    int s1_how, contiguous_how, strided_how; bool s1_fused, mv_fused;
    explicit SumOrder(int tie_mode)
    {
        switch (tie_mode >> 4) {
        default: case 0: s1_how = 0; contiguous_how = 0; strided_how = 0; s1_fused = true; mv_fused = false; break;
        case 1: s1_how = 0; contiguous_how = 0; strided_how = 0; s1_fused = false; mv_fused = false; break;
        case 2: s1_how = 0; contiguous_how = 4; strided_how = 0; s1_fused = false; mv_fused = false; break;
        case 3: s1_how = 4; contiguous_how = 4; strided_how = 4; s1_fused = false; mv_fused = false; break;
        case 4: s1_how = 8; contiguous_how = 8; strided_how = 8; s1_fused = true; mv_fused = true; break;
        case 5: s1_how = -1; contiguous_how = -1; strided_how = -1; s1_fused = false; mv_fused = false; break;
        }
    }
};

// ---- byte reader with ifstream-like semantics -------------------------------------------------
struct Reader {
    const uint8_t* p; size_t len; size_t pos = 0; bool fail = false;
    // ifstream::read past EOF delivers what is left and then fails every later read
    // (matcher.cpp:975 relies on this for the rolled codes).  Bytes not delivered are zero here
    // (the reference leaves stack garbage).
    void read(void* dst, size_t n)
    {
        if (fail) { return; }
        size_t avail = len - pos;
        if (n > avail) { memcpy(dst, p + pos, avail); pos = len; fail = true; return; }
        memcpy(dst, p + pos, n); pos += n;
    }
};

// ---- F2: codebook ------------------------------------------------------------------------------
bool load_codebook_bytes(const uint8_t* buf, size_t len, Codebook& cb)
{
    // matcher.cpp:45-56
    cb.dist_N = 50; cb.N = 200;
    cb.table_dist.assign(cb.dist_N * cb.dist_N, 0.f);
    for (int i = 0; i < cb.dist_N; ++i)
        for (int j = i; j < cb.dist_N; ++j) {
            cb.table_dist[i * cb.dist_N + j] = (float)sqrt((i * 16.0) * (i * 16.0) + (j * 16.0) * (j * 16.0));
            cb.table_dist[j * cb.dist_N + i] = cb.table_dist[i * cb.dist_N + j];
        }
    // matcher.cpp:74-93
    Reader r{buf, len};
    short a = 0, b = 0, c = 0;
    r.read(&a, 2); r.read(&b, 2); r.read(&c, 2);
    cb.M = a; cb.K = b; cb.dsub = c;
    long n = (long)cb.M * cb.K * cb.dsub;
    if (n <= 0 || r.fail) return false;
    cb.cw.assign(n, 0.f);
    r.read(cb.cw.data(), sizeof(float) * n);
    return !r.fail;
}

// ---- S4: per-query ADC look-up table -------------------------------------------------------------
// include.h:327-359.  lut[i][j][q] = sum_{k<dsub} (des[i][j*dsub+k] - cw[j][q][k])^2, k ascending,
// float arithmetic, product and sum rounded separately.
void build_lut(const float* des, int n, int des_len, const Codebook& cb, std::vector<float>& lut)
{
    lut.assign((size_t)n * cb.M * cb.K, 0.f);
    for (int i = 0; i < n; ++i) {
        const float* pdes0 = des + (size_t)i * des_len;
        for (int j = 0; j < cb.M; ++j) {
            const float* pdes1 = pdes0 + j * cb.dsub;
            const float* pword0 = cb.cw.data() + (size_t)j * cb.K * cb.dsub;
            for (int q = 0; q < cb.K; ++q) {
                const float* pword1 = pword0 + q * cb.dsub;
                float dist = 0.0f;
                for (int k = 0; k < cb.dsub; ++k) {
                    float d = pdes1[k] - pword1[k];
                    float d2 = d * d;
                    dist += d2;
                }
                lut[(size_t)i * cb.M * cb.K + j * cb.K + q] = dist;
            }
        }
    }
}

// ---- F1: template .dat parsing -------------------------------------------------------------------
// matcher.cpp:785-884 (latent) and :886-983 (rolled).  Return codes as the reference: 0 ok, 1 empty
// file, 2 too many minutiae in a minutiae template, 4 ridge-flow block too large, -1 texture
// template too large.  Templates with n <= 0 are skipped (later indices shift, :835-836,:863-864).
const int kMaxMinu = 2000, kMaxDesLen = 192, kMaxBlk = 100;

static void read_points(Reader& r, int n, std::vector<Point>& pts)
{
    std::vector<short> x(n), y(n); std::vector<float> o(n);
    r.read(x.data(), 2 * (size_t)n); r.read(y.data(), 2 * (size_t)n); r.read(o.data(), 4 * (size_t)n);
    pts.resize(n);
    for (int i = 0; i < n; ++i) pts[i] = Point{x[i], y[i], o[i]};  // short -> int, include.h:171-193
}

int parse_latent(const uint8_t* buf, size_t len, const Codebook& cb, Latent& L)
{
    L.minu.clear(); L.tex.clear();
    if (len == 0) return 1;                                   // :798-801
    Reader r{buf, len};
    short header[12]; r.read(header, 24);
    short h = 0, w = 0, blkH = 0, blkW = 0; unsigned char nmt = 0, ntt = 0;
    r.read(&h, 2); r.read(&w, 2); r.read(&blkH, 2); r.read(&blkW, 2); r.read(&nmt, 1);
    if (blkH > 50) blkH = 50;
    if (blkW > 50) blkW = 50;
    if (r.fail) nmt = 0;
    for (int i = 0; i < nmt; ++i) {
        short n = 0; r.read(&n, 2);
        if (r.fail) break;
        if (n <= 0) continue;
        if (n > kMaxMinu) return 2;
        if (blkH > kMaxBlk || blkW > kMaxBlk) return 4;
        MinuTpl t; t.n = n;
        read_points(r, n, t.pts);
        short dl = 0; r.read(&dl, 2);
        if (r.fail || dl <= 0 || dl > kMaxDesLen) break;
        t.des_len = dl; t.des.assign((size_t)n * dl, 0.f);
        r.read(t.des.data(), 4 * (size_t)n * dl);
        L.minu.push_back(std::move(t));
    }
    r.read(&ntt, 1);
    if (r.fail) ntt = 0;
    for (int i = 0; i < ntt; ++i) {
        short n = 0; r.read(&n, 2);
        if (r.fail) break;
        if (n <= 0) continue;
        if (n > kMaxMinu) return -1;
        LatTexTpl t; t.n = n;
        read_points(r, n, t.pts);
        short dl = 0; r.read(&dl, 2);
        if (r.fail || dl <= 0 || dl > kMaxDesLen) break;
        t.des_len = dl; t.des.assign((size_t)n * dl, 0.f);
        r.read(t.des.data(), 4 * (size_t)n * dl);
        build_lut(t.des.data(), t.n, t.des_len, cb, t.lut);     // :878
        L.tex.push_back(std::move(t));
    }
    return 0;
}

int parse_rolled(const uint8_t* buf, size_t len, Rolled& R)
{
    R.minu.clear(); R.tex.clear();
    if (len <= 10) return 1;                                   // :899-902
    Reader r{buf, len};
    short header[12]; r.read(header, 24);
    short h = 0, w = 0, blkH = 0, blkW = 0; unsigned char nmt = 0, ntt = 0;
    r.read(&h, 2); r.read(&w, 2); r.read(&blkH, 2); r.read(&blkW, 2); r.read(&nmt, 1);
    if (blkH > 50) blkH = 50;
    if (blkW > 50) blkW = 50;
    if (r.fail) nmt = 0;
    for (int i = 0; i < nmt; ++i) {
        short n = 0; r.read(&n, 2);
        if (r.fail) break;
        if (n <= 0) continue;
        if (n > kMaxMinu) return 2;
        if (blkH > kMaxBlk || blkW > kMaxBlk) return 4;
        MinuTpl t; t.n = n;
        read_points(r, n, t.pts);
        short dl = 0; r.read(&dl, 2);
        if (r.fail || dl <= 0 || dl > kMaxDesLen) break;
        t.des_len = dl; t.des.assign((size_t)n * dl, 0.f);
        r.read(t.des.data(), 4 * (size_t)n * dl);
        R.minu.push_back(std::move(t));
    }
    r.read(&ntt, 1);
    if (r.fail) ntt = 0;
    for (int i = 0; i < ntt; ++i) {
        short n = 0; r.read(&n, 2);
        if (r.fail) break;
        if (n <= 0) continue;
        if (n > kMaxMinu) return -1;
        RolTexTpl t; t.n = n;
        read_points(r, n, t.pts);
        short dl = 0; r.read(&dl, 2);
        if (r.fail || dl <= 0 || dl > kMaxDesLen) break;
        t.des_len = dl;
        // :975 reads n*des_len FLOATS (4x over-read, relies on EOF) and include.h:401-406 keeps the
        // first n*des_len BYTES as the PQ codes.
        std::vector<uint8_t> over((size_t)n * dl * 4, 0);
        r.read(over.data(), over.size());
        t.codes.assign(over.begin(), over.begin() + (size_t)n * dl);
        R.tex.push_back(std::move(t));
    }
    return 0;
}

// ---- S9 helper -------------------------------------------------------------------------------------
// matcher.cpp:1638-1647: float in/out, comparisons and the +-2*PI in double.
static inline float adjust_angle(float angle)
{
    if (angle > ORC_PI) angle -= 2 * ORC_PI;
    else if (angle < -ORC_PI) angle += 2 * ORC_PI;
    return angle;
}

// ---- greedy selection shared by S8a/S8b/S9 -----------------------------------------------------------
// matcher.cpp:1304-1344 / :1425-1465 / :1593-1633.  `compatible(a,b)` is H[a][b] >= 1e-5 (S8) or H[a][b]
// true (S9); `s_thr` is 0.0001 (S8) or 0.001 (S9), compared in double.
template <class Compat>
std::vector<Corr> greedy_select(const std::vector<Corr>& corr, const std::vector<float>& S, int nL, int nR,
                                double s_thr, Compat compatible, int tie_mode, int site)
{
    int num = (int)corr.size();
    std::vector<int> y(num);
    std::iota(y.begin(), y.end(), 0);
    sort_idx(y, [&S](int a, int b) { return S[a] > S[b]; }, tie_mode, site);
    std::vector<short> flag_latent(nL, 0), flag_rolled(nR, 0);
    std::vector<Corr> out; std::vector<int> sel;
    for (int i = 0; i < num; ++i) {
        short ind = (short)y[i];
        if (S[ind] < s_thr) break;
        if (flag_latent[std::get<1>(corr[ind])] == 1 | flag_rolled[std::get<2>(corr[ind])] == 1) continue;
        bool found = false;
        if (i != 0) {
            for (size_t j = 0; j < sel.size(); ++j)
                if (!compatible(ind, sel[j])) { found = true; break; }
        }
        if (!found) {
            sel.push_back(ind);
            out.push_back(corr[ind]);
            flag_latent[std::get<1>(corr[ind])] = 1;
            flag_rolled[std::get<2>(corr[ind])] = 1;
        }
    }
    return out;
}

// ---- S8a / S8b: distance-consistency graph + power iteration ---------------------------------------------
// S8a = LSS_R_Fast2_Dist_eigen  matcher.cpp:1350-1469 (pixel coords, sqrtf, 5 iterations)
// S8b = LSS_R_Fast2_Dist_lookup matcher.cpp:1225-1348 (block coords, table_dist, 3 iterations)
std::vector<Corr> dist_filter(const std::vector<Corr>& corr, const std::vector<Point>& Lp, const std::vector<Point>& Rp,
                              const Codebook& cb, bool lookup, int iters, int tie_mode)
{
    const float d_thr = 30.0f;                 // called with int 30 -> float parameter (:491-492,:758-759)
    int num = (int)corr.size();
    std::vector<float> H((size_t)num * num, 0.f);
    for (int i = 0; i < num - 1; ++i) {
        const Point& l1 = Lp[std::get<1>(corr[i])];
        const Point& r1 = Rp[std::get<2>(corr[i])];
        for (int j = i + 1; j < num; ++j) {
            const Point& l2 = Lp[std::get<1>(corr[j])];
            const Point& r2 = Rp[std::get<2>(corr[j])];
            float dist_1, dist_2;
            if (lookup) {
                int dx_1 = abs(l1.x - l2.x), dx_2 = abs(r1.x - r2.x);
                int dy_1 = abs(l1.y - l2.y), dy_2 = abs(r1.y - r2.y);
                if (dx_1 >= cb.dist_N | dx_2 >= cb.dist_N | dy_1 >= cb.dist_N | dy_2 >= cb.dist_N) continue;  // :1257
                dist_1 = cb.table_dist[dx_1 * cb.dist_N + dy_1];
                dist_2 = cb.table_dist[dx_2 * cb.dist_N + dy_2];
            } else {
                float dx_1 = (float)(l1.x - l2.x), dx_2 = (float)(r1.x - r2.x);
                float dy_1 = (float)(l1.y - l2.y), dy_2 = (float)(r1.y - r2.y);
                float a = dx_1 * dx_1, b = dy_1 * dy_1;
                dist_1 = a + b; dist_1 = sqrtf(dist_1);           // :1380-1381 (sqrt(float) -> float)
                float c = dx_2 * dx_2, d = dy_2 * dy_2;
                dist_2 = c + d; dist_2 = sqrtf(dist_2);
            }
            float dist = fabsf(dist_1 - dist_2);
            if (dist > d_thr) continue;
            float hv = (float)((double)(30 - dist) / (25.0));       // :1268,:1389 float-30 -> double divide -> float
            if (hv > 1) hv = 1.0f; else if (hv < 0) hv = 0.0f;
            H[(size_t)i * num + j] = hv;
            H[(size_t)j * num + i] = hv;
        }
    }
    // power iteration, :1279-1289 / :1401-1411.  Canonical order (Eigen's is unpinned): c_j = sum_k H[j][k]*b[k]
    // k ascending, unfused; sum = sum_j c_j ascending; the 1/(sum+1e-5) factor is computed in double and
    // narrowed to float before the multiply (Eigen narrows a double scalar to the vector's scalar type).
    const SumOrder so(tie_mode);
    std::vector<float> b(num), c(num);
    for (int i = 0; i < num; ++i) b[i] = std::get<0>(corr[i]);
    for (int it = 0; it < iters; ++it) {
        float sum = 0.0f;
        for (int j = 0; j < num; ++j) c[j] = dot_how(&H[(size_t)j * num], b.data(), num, so.contiguous_how, so.mv_fused);
        sum = sum_how(c.data(), num, 1, so.contiguous_how);
        float scale = (float)(1. / (sum + 0.00001));
        for (int j = 0; j < num; ++j) b[j] = c[j] * scale;
    }
    auto compat = [&H, num](int a, int bb) { return !((double)H[(size_t)a * num + bb] < 0.00001); };
    return greedy_select(corr, b, (int)Lp.size(), (int)Rp.size(), 0.0001, compat, tie_mode, kSiteS8);
}

// ---- S9: angle-consistency graph -----------------------------------------------------------------------------
// LSS_R_Fast2, matcher.cpp:1471-1636.
std::vector<Corr> angle_filter(const std::vector<Corr>& corr, const std::vector<Point>& Lp, const std::vector<Point>& Rp, int tie_mode)
{
    int num = (int)corr.size();
    std::vector<char> H((size_t)num * num, 0);
    for (int i = 0; i < num - 1; ++i) {
        const Point& l1 = Lp[std::get<1>(corr[i])];
        const Point& r1 = Rp[std::get<2>(corr[i])];
        for (int j = i + 1; j < num; ++j) {
            const Point& l2 = Lp[std::get<1>(corr[j])];
            const Point& r2 = Rp[std::get<2>(corr[j])];
            float angle_1 = l1.ori - l2.ori; angle_1 = adjust_angle(angle_1);
            float angle_2 = r1.ori - r2.ori; angle_2 = adjust_angle(angle_2);
            float angle_diff = fabsf(angle_1 - angle_2);
            if (angle_diff > ORC_PI) angle_diff = 2 * ORC_PI - angle_diff;
            if (angle_diff > ORC_PI / 4.) continue;

            float dx_1 = (float)(l1.x - l2.x), dy_1 = (float)(l1.y - l2.y);
            float line_angle_1 = -atan2f(dy_1, dx_1);                  // :1516 atan2(float,float) -> float
            angle_1 = l1.ori - line_angle_1; angle_1 = adjust_angle(angle_1);
            float dx_2 = (float)(r1.x - r2.x), dy_2 = (float)(r1.y - r2.y);
            float line_angle_2 = -atan2f(dy_2, dx_2);
            angle_2 = r1.ori - line_angle_2; angle_2 = adjust_angle(angle_2);
            angle_diff = fabsf(angle_1 - angle_2);
            if (angle_diff > ORC_PI) angle_diff = 2 * ORC_PI - angle_diff;
            if (angle_diff > ORC_PI / 6.) continue;

            angle_1 = l2.ori - line_angle_1; angle_1 = adjust_angle(angle_1);
            angle_2 = r2.ori - line_angle_2; angle_2 = adjust_angle(angle_2);
            angle_diff = fabsf(angle_1 - angle_2);
            if (angle_diff > ORC_PI) angle_diff = 2 * ORC_PI - angle_diff;
            if (angle_diff > ORC_PI / 6.) continue;

            H[(size_t)i * num + j] = 1;
            H[(size_t)j * num + i] = 1;
        }
    }
    std::vector<float> S(num), S1(num);
    float s0 = (float)(1.0 / num);                                   // :1558
    for (int i = 0; i < num; ++i) S[i] = s0;
    for (int it = 0; it < 5; ++it) {                                   // :1563-1581 (the reference's own loops)
        float sum = 0.0f;
        for (int j = 0; j < num; ++j) {
            S1[j] = 0;
            for (int k = 0; k < num; ++k)
                if (H[(size_t)j * num + k]) S1[j] += S[k];
            sum += S1[j];
        }
        sum = (float)(1.0 / (sum + 0.00001));
        for (int j = 0; j < num; ++j) S[j] = S1[j] * sum;
    }
    auto compat = [&H, num](int a, int b) { return H[(size_t)a * num + b] != 0; };
    return greedy_select(corr, S, (int)Lp.size(), (int)Rp.size(), 0.001, compat, tie_mode, kSiteS9);
}

static float sum_scores(const std::vector<Corr>& c)   // :508-514, :775-781
{
    float score = 0.0f;
    for (size_t i = 0; i < c.size(); ++i) score += std::get<0>(c[i]);
    return score;
}

struct MinuTrace { std::vector<Corr> corr, corr2, corr3; std::vector<float> simi, norm; };
struct TexTrace { std::vector<float> rowmax; std::vector<int> rowarg; std::vector<Corr> corr, corr2, corr3; bool rowmax_only = false; };   // rowmax_only: stop after S6 (the tap of S5 + S6; also keeps NaN row maxima away from the sorts of S7)

// ---- S1-S3 (+S8a, S9): minutiae-template scorer -----------------------------------------------------------------
// One2One_minutiae_matching, matcher.cpp:420-516.
float minutiae_score(const MinuTpl& L, const MinuTpl& R, const Codebook& cb, int tie_mode, MinuTrace* tr)
{
    int nL = L.n, nR = R.n, D = R.des_len;
    if (D != L.des_len) return NAN;                                    // assert at :433
    std::vector<float> simi((size_t)nL * nR);
    const SumOrder so(tie_mode);
    // S1 (:440-452).  Canonical order: fmaf chain, k ascending (Eigen's order is unpinned).
    for (int i = 0; i < nL; ++i)
        for (int j = 0; j < nR; ++j) {
            const float* a = &L.des[(size_t)i * D]; const float* b = &R.des[(size_t)j * D];
            float acc = dot_how(a, b, D, so.s1_how, so.s1_fused);
            if (acc < 0) acc = 0;
            simi[(size_t)i * nR + j] = acc;
        }
    // S2 (:455-470): column sums (rolled) and row sums (latent), index ascending.
    std::vector<float> rs(nR, 0.f), ls(nL, 0.f);
    for (int j = 0; j < nR; ++j) rs[j] = sum_how(&simi[j], nL, nR, so.strided_how);
    for (int i = 0; i < nL; ++i) ls[i] = sum_how(&simi[(size_t)i * nR], nR, 1, so.contiguous_how);
    std::vector<float> norm((size_t)nL * nR);
    for (int i = 0; i < nL; ++i)
        for (int j = 0; j < nR; ++j) {
            float s = simi[(size_t)i * nR + j];
            float f = ls[i] + rs[j];
            f = f - s;
            norm[(size_t)i * nR + j] = (float)((double)s / ((double)f + 0.000001));   // :467
        }
    // S3 (:473-488)
    std::vector<int> y((size_t)nL * nR);
    std::iota(y.begin(), y.end(), 0);
    sort_idx(y, [&norm](int a, int b) { return norm[a] > norm[b]; }, tie_mode, kSiteS3);
    int topN = 120;
    if (nR * nL < topN) topN = nR * nL;
    std::vector<Corr> corr;
    for (int i = 0; i < topN; ++i) {
        int i1 = y[i] / nR, i2 = y[i] - i1 * nR;
        corr.push_back(std::make_tuple(simi[(size_t)i1 * nR + i2], i1, i2));
    }
    std::vector<Corr> corr2 = dist_filter(corr, L.pts, R.pts, cb, false, 5, tie_mode);      // :492
    std::vector<Corr> corr3 = angle_filter(corr2, L.pts, R.pts, tie_mode);                   // :495
    if (tr) { tr->corr = corr; tr->corr2 = corr2; tr->corr3 = corr3; tr->simi = simi; tr->norm = norm; }
    return sum_scores(corr3);
}

// ---- S5-S7 (+S8b, S9): texture-template scorer --------------------------------------------------------------------
// One2One_texture_matching, matcher.cpp:531-783 (method 1).
float texture_score(const LatTexTpl& L, const RolTexTpl& R, const Codebook& cb, int tie_mode, TexTrace* tr)
{
    int nL = std::min(L.n, 1000), nR = std::min(R.n, 1000);           // :544-547
    int K = cb.K, M = cb.M;
    std::vector<float> rowmax(nL); std::vector<int> rowarg(nL);
    for (int i = 0; i < nL; ++i) {
        const float* lut = &L.lut[(size_t)i * M * K];
        float best = 0.f; int besti = 0;
        for (int j = 0; j < nR; ++j) {
            const uint8_t* c = &R.codes[(size_t)j * R.des_len];
            float d1 = 6.f, d2 = 0.f, d3 = 0.f, d4 = 0.f;                // :571-574
            for (int k = 0; k < M; k += 4) {                            // :577-591
                d1 -= lut[(k + 0) * K + c[k + 0]];
                d2 -= lut[(k + 1) * K + c[k + 1]];
                d3 -= lut[(k + 2) * K + c[k + 2]];
                d4 -= lut[(k + 3) * K + c[k + 3]];
            }
            float s = (d1 + d2) + (d3 + d4);                              // :592
            if (j == 0 || s > best) { best = s; besti = j; }              // std::max_element: first maximum (:730)
        }
        rowmax[i] = best; rowarg[i] = besti;
    }
    if (tr && tr->rowmax_only) { tr->rowmax = rowmax; tr->rowarg = rowarg; return 0.f; }
    std::vector<Corr> tmp(nL), corr;
    for (int i = 0; i < nL; ++i) tmp[i] = std::make_tuple(rowmax[i], i, rowarg[i]);
    if ((int)tmp.size() > cb.N) {                                         // :736-747
        std::vector<int> y(tmp.size());
        std::iota(y.begin(), y.end(), 0);
        sort_idx(y, [&tmp](int a, int b) { return std::get<0>(tmp[a]) > std::get<0>(tmp[b]); }, tie_mode, kSiteS7);
        corr.resize(cb.N);
        for (int i = 0; i < cb.N; ++i) corr[i] = tmp[y[i]];
    } else corr = tmp;                                                     // :748-749
    // the filters index flags by m_nrof_minu (already clamped) and points by the correspondence indices
    std::vector<Corr> corr2 = dist_filter(corr, L.pts, R.pts, cb, true, 3, tie_mode);       // :759
    std::vector<Corr> corr3 = angle_filter(corr2, L.pts, R.pts, tie_mode);                   // :767
    if (tr) { tr->rowmax = rowmax; tr->rowarg = rowarg; tr->corr = corr; tr->corr2 = corr2; tr->corr3 = corr3; }
    return sum_scores(corr3);
}

// ---- S10: template selection + fusion ------------------------------------------------------------------------------
// One2One_matching_selected_templates matcher.cpp:376-417 and the fusion at :188 / :293.
// out[0..2] minutiae scores (latent templates 26, 2, 11), out[3] texture score, out[4] final.
// Returns 0 ok, 1 latent empty, 2 rolled empty.  The reference reads score[28] (and score[0..2]) out of
// bounds when the latent has fewer than 29 (3) score slots; here an absent slot reads as 0.
int pair_score(const Latent& L, const Rolled& R, const Codebook& cb, int tie_mode, float out[5])
{
    for (int i = 0; i < 5; ++i) out[i] = 0.f;
    int nLm = (int)L.minu.size(), nLt = (int)L.tex.size(), nRm = (int)R.minu.size(), nRt = (int)R.tex.size();
    const int sel[3] = {27 - 1, 3 - 1, 12 - 1};
    if (nLm <= sel[0] && nLt <= 0) return 1;
    if (nRm <= 0 && nRt <= 0) return 2;
    std::vector<float> score(nLm + nLt, 0.f);
    for (int i = 0; i < 3 && nRm > 0; ++i) {
        int ind = sel[i];
        if (nLm <= ind) continue;
        float s = minutiae_score(L.minu[ind], R.minu[0], cb, tie_mode, nullptr);
        score[i] = s; out[i] = s;
    }
    for (int i = 0; i < std::min(1, nLt) && nRt > 0; ++i) {
        float s = texture_score(L.tex[i], R.tex[0], cb, tie_mode, nullptr);
        score[i + nLm] = s; out[3] = s;
    }
    auto at = [&score](size_t k) { return k < score.size() ? score[k] : 0.f; };
    float f = at(0) + at(1);
    f = f + at(2);
    out[4] = (float)((double)f + (double)at(28) * 0.3);                   // :188
    return 0;
}

// One2One_matching_all_templates, matcher.cpp:339-374: score[i] = minutiae scorer of latent template i vs rolled template 0 for
// every i (while the rolled print has a minutiae template), score[nLm + t] = texture scorer of latent texture template t vs
// rolled texture template 0; 1 = latent without any template, 2 = rolled without any template (vector stays zero).
int all_templates_score(const Latent& L, const Rolled& R, const Codebook& cb, int tie_mode, std::vector<float>& score)
{
    int nLm = (int)L.minu.size(), nLt = (int)L.tex.size(), nRm = (int)R.minu.size(), nRt = (int)R.tex.size();
    score.assign(nLm + nLt, 0.f);
    if (nLm <= 0 && nLt <= 0) return 1;
    if (nRm <= 0 && nRt <= 0) return 2;
    for (int i = 0; i < nLm && nRm; ++i) score[i] = minutiae_score(L.minu[i], R.minu[0], cb, tie_mode, nullptr);
    for (int i = 0; i < nLt && nRt > 0; ++i) score[i + nLm] = texture_score(L.tex[i], R.tex[0], cb, tie_mode, nullptr);
    return 0;
}

static bool read_file(const char* path, std::vector<uint8_t>& buf)
{
    std::ifstream is(path, std::ifstream::binary);
    if (!is) { buf.clear(); return false; }
    is.seekg(0, std::ios::end); long n = is.tellg(); is.seekg(0, std::ios::beg);
    buf.resize(n > 0 ? n : 0);
    if (n > 0) is.read((char*)buf.data(), n);
    return true;
}

}  // namespace

// =========================================== C ABI for ctypes ===========================================
extern "C" {

void* orc_codebook_from_bytes(const uint8_t* buf, long len)
{
    Codebook* cb = new Codebook();
    if (!load_codebook_bytes(buf, (size_t)len, *cb)) { delete cb; return nullptr; }
    return cb;
}
void* orc_codebook_load(const char* path)
{
    std::vector<uint8_t> b;
    if (!read_file(path, b)) return nullptr;
    return orc_codebook_from_bytes(b.data(), (long)b.size());
}
void orc_codebook_free(void* cb) { delete (Codebook*)cb; }
void orc_codebook_dims(void* cb, int* M, int* K, int* dsub) { Codebook* c = (Codebook*)cb; *M = c->M; *K = c->K; *dsub = c->dsub; }
const float* orc_codebook_words(void* cb) { return ((Codebook*)cb)->cw.data(); }
const float* orc_codebook_table_dist(void* cb) { return ((Codebook*)cb)->table_dist.data(); }

void* orc_latent_from_bytes(void* cb, const uint8_t* buf, long len, int* rc)
{
    Latent* L = new Latent();
    L->load_rc = parse_latent(buf, (size_t)len, *(Codebook*)cb, *L);
    if (rc) *rc = L->load_rc;
    return L;
}
void* orc_rolled_from_bytes(const uint8_t* buf, long len, int* rc)
{
    Rolled* R = new Rolled();
    R->load_rc = parse_rolled(buf, (size_t)len, *R);
    if (R->load_rc < 0) { R->minu.clear(); R->tex.clear(); }            // matcher.cpp:173-177
    if (rc) *rc = R->load_rc;
    return R;
}
void* orc_latent_load(void* cb, const char* path, int* rc)
{
    std::vector<uint8_t> b; read_file(path, b);
    return orc_latent_from_bytes(cb, b.data(), (long)b.size(), rc);
}
void* orc_rolled_load(const char* path, int* rc)
{
    std::vector<uint8_t> b; read_file(path, b);
    return orc_rolled_from_bytes(b.data(), (long)b.size(), rc);
}
void orc_latent_free(void* p) { delete (Latent*)p; }
void orc_rolled_free(void* p) { delete (Rolled*)p; }

// shape queries: counts[0]=n minutiae templates, counts[1]=n texture templates
void orc_latent_counts(void* p, int* counts) { Latent* L = (Latent*)p; counts[0] = (int)L->minu.size(); counts[1] = (int)L->tex.size(); }
void orc_rolled_counts(void* p, int* counts) { Rolled* R = (Rolled*)p; counts[0] = (int)R->minu.size(); counts[1] = (int)R->tex.size(); }
int orc_latent_minu_n(void* p, int t) { Latent* L = (Latent*)p; return t < (int)L->minu.size() ? L->minu[t].n : -1; }
int orc_latent_tex_n(void* p, int t) { Latent* L = (Latent*)p; return t < (int)L->tex.size() ? L->tex[t].n : -1; }
int orc_rolled_minu_n(void* p, int t) { Rolled* R = (Rolled*)p; return t < (int)R->minu.size() ? R->minu[t].n : -1; }
int orc_rolled_tex_n(void* p, int t) { Rolled* R = (Rolled*)p; return t < (int)R->tex.size() ? R->tex[t].n : -1; }
const float* orc_latent_lut(void* p, int t) { Latent* L = (Latent*)p; return L->tex[t].lut.data(); }
const uint8_t* orc_rolled_codes(void* p, int t) { Rolled* R = (Rolled*)p; return R->tex[t].codes.data(); }
// copy points of a template: kind 0 latent-minutiae, 1 latent-texture, 2 rolled-minutiae, 3 rolled-texture
int orc_points(void* p, int kind, int t, int* x, int* y, float* ori)
{
    const std::vector<Point>* pts = nullptr;
    if (kind == 0) pts = &((Latent*)p)->minu[t].pts;
    else if (kind == 1) pts = &((Latent*)p)->tex[t].pts;
    else if (kind == 2) pts = &((Rolled*)p)->minu[t].pts;
    else pts = &((Rolled*)p)->tex[t].pts;
    for (size_t i = 0; i < pts->size(); ++i) { x[i] = (*pts)[i].x; y[i] = (*pts)[i].y; ori[i] = (*pts)[i].ori; }
    return (int)pts->size();
}

// S4 standalone
void orc_build_lut(void* cb, const float* des, int n, int des_len, float* out)
{
    std::vector<float> lut;
    build_lut(des, n, des_len, *(Codebook*)cb, lut);
    memcpy(out, lut.data(), lut.size() * sizeof(float));
}

// PQ encoder: TrainedPQEncoder.encode_multi, extraction/descriptor_PQ.py:19-27 — per sub-space the index of the
// nearest codeword, computed by scipy.cluster.vq.vq (third-party, version unpinned by the reference).  Restated as: squared L2
// in fp32 with the SAME arithmetic as the matcher's own table (include.h:327-359, build_lut above), first minimum on ties — so
// a point's code is the codeword its own ADC table ranks nearest.  Pinned against scipy 1.15.3's vq (float32 path) run in the
// build container: tests/golden/golden_pq.npz, 0 differences on 8192 (point, sub-space) cases (tests/test_oracle.py).
void orc_pq_encode(void* cbp, const float* des, int n, int des_len, unsigned char* codes)
{
    const Codebook& cb = *(Codebook*)cbp;
    std::vector<float> lut;
    for (int i = 0; i < n; ++i) {
        build_lut(des + (size_t)i * des_len, 1, des_len, cb, lut);
        for (int j = 0; j < cb.M; ++j) {
            const float* row = lut.data() + (size_t)j * cb.K;
            int best = 0;
            for (int q = 1; q < cb.K; ++q) if (row[q] < row[best]) best = q;
            codes[(size_t)i * cb.M + j] = (unsigned char)best;
        }
    }
}

// matcher.cpp:339-374; out capacity >= n_minu + n_tex of the latent; returns the reference's code (0, 1, 2)
int orc_all_templates(void* cb, void* lat, void* rol, int tie_mode, float* out)
{
    std::vector<float> sc;
    int rc = all_templates_score(*(Latent*)lat, *(Rolled*)rol, *(Codebook*)cb, tie_mode, sc);
    memcpy(out, sc.data(), sc.size() * sizeof(float));
    return rc;
}

// S10: per-pair scores out[5] = s0,s1,s2,tex,final
int orc_pair_score(void* cb, void* lat, void* rol, int tie_mode, float* out)
{
    return pair_score(*(Latent*)lat, *(Rolled*)rol, *(Codebook*)cb, tie_mode, out);
}

// S5/S6 intermediates for latent texture template 0 vs rolled texture template 0
int orc_texture_rowmax(void* cb, void* lat, void* rol, float* val, int* arg)
{
    Latent* L = (Latent*)lat; Rolled* R = (Rolled*)rol;
    if (L->tex.empty() || R->tex.empty()) return 0;
    TexTrace tr; tr.rowmax_only = true;
    texture_score(L->tex[0], R->tex[0], *(Codebook*)cb, 1, &tr);
    memcpy(val, tr.rowmax.data(), tr.rowmax.size() * 4);
    memcpy(arg, tr.rowarg.data(), tr.rowarg.size() * 4);
    return (int)tr.rowmax.size();
}

// Stage traces.  which: 0 = texture scorer (tex 0 vs tex 0), 1..3 = minutiae scorer for selected template
// 26/2/11 vs rolled 0.  stage: 0 corr (after S3/S7), 1 after S8, 2 after S9.  Returns the count; fills
// sim/li/ri (capacity >= 200).
int orc_trace(void* cbp, void* lat, void* rol, int tie_mode, int which, int stage, float* sim, int* li, int* ri)
{
    Latent* L = (Latent*)lat; Rolled* R = (Rolled*)rol; Codebook* cb = (Codebook*)cbp;
    std::vector<Corr> c;
    if (which == 0) {
        if (L->tex.empty() || R->tex.empty()) return -1;
        TexTrace tr; texture_score(L->tex[0], R->tex[0], *cb, tie_mode, &tr);
        c = stage == 0 ? tr.corr : stage == 1 ? tr.corr2 : tr.corr3;
    } else {
        const int sel[3] = {26, 2, 11};
        int ind = sel[which - 1];
        if ((int)L->minu.size() <= ind || R->minu.empty()) return -1;
        MinuTrace tr; minutiae_score(L->minu[ind], R->minu[0], *cb, tie_mode, &tr);
        c = stage == 0 ? tr.corr : stage == 1 ? tr.corr2 : tr.corr3;
    }
    for (size_t i = 0; i < c.size(); ++i) { sim[i] = std::get<0>(c[i]); li[i] = std::get<1>(c[i]); ri[i] = std::get<2>(c[i]); }
    return (int)c.size();
}

// The libm atan2f this oracle (and a CPU build of the reference, matcher.cpp:1516/:1524) evaluates, on the grid of integer
// coordinate differences: out[(dy + R) * (2R + 1) + (dx + R)] = atan2f((float)dy, (float)dx).  The GPU tests compare the device's
// atan2 against this table exhaustively over the coordinate range of real templates.
void orc_atan2f_grid(int R, float* out)
{
    const int W = 2 * R + 1;
#pragma omp parallel for
    for (int dy = -R; dy <= R; ++dy)
        for (int dx = -R; dx <= R; ++dx) out[(size_t)(dy + R) * W + (dx + R)] = atan2f((float)dy, (float)dx);
}

// S11's rank list, matcher.cpp:306-309: the gallery indices sorted by score, descending, non-strict comparator.  std_sort != 0: with std::sort as the reference (equal scores in
// whatever order libstdc++'s introsort leaves them — what `match -l ... -tie 1|2` writes); 0: equal scores by ascending index (afis_search's top-k, the CLI's default).
void orc_rank_list(const float* scores, int n, int std_sort, int* out)
{
    std::vector<int> ind(n);
    std::iota(ind.begin(), ind.end(), 0);
    auto cmp = [scores](const int& a, const int& b) { return scores[a] > scores[b]; };
    if (std_sort) std::sort(ind.begin(), ind.end(), cmp); else std::stable_sort(ind.begin(), ind.end(), cmp);
    std::copy(ind.begin(), ind.end(), out);
}

// S11: one latent against a list of rolled handles (the body of the OpenMP loop, matcher.cpp:168-190).
// scores[j] = final or -1 (rolled empty).  Returns 1 if the latent is empty (whole query skipped).
// threads <= 0: the reference's own setting, 8 threads schedule(static,16).
int orc_search(void* cb, void* lat, void** rolled, int n, int tie_mode, int threads, float* scores, float* parts /*[n][5] or NULL*/)
{
    int result = 0;
    int nt = threads <= 0 ? 8 : threads;
    (void)nt;
    // threads <= 0: the reference's pragma (8 threads, static chunks of 16: n/16 chunks bound the threads that get any work).
    // threads > 0 (the "best the host can do" legs of bench.py): one pair at a time to whichever thread is free.
#ifdef _OPENMP
    omp_set_schedule(threads <= 0 ? omp_sched_static : omp_sched_dynamic, threads <= 0 ? 16 : 1);
#endif
#pragma omp parallel for num_threads(nt) schedule(runtime)
    for (int j = 0; j < n; ++j) {
        float out[5];
        scores[j] = -1.f;
        int rc = pair_score(*(Latent*)lat, *(Rolled*)rolled[j], *(Codebook*)cb, tie_mode, out);
        if (parts) memcpy(parts + (size_t)j * 5, out, sizeof(out));
        if (rc == 1) { result = 1; continue; }
        if (rc == 2) continue;
        scores[j] = out[4];
    }
    return result;
}

// S11 exactly as the reference runs it (matcher.cpp:168-190 / :273-295): the same loop, but every rolled template is re-read from
// its file and re-parsed for every (latent, rolled) pair (load_FP_template inside the loop, :173 / :278).  threads <= 0: 8.
int orc_search_files(void* cb, void* lat, const char* const* paths, int n, int tie_mode, int threads, float* scores)
{
    int result = 0;
    int nt = threads <= 0 ? 8 : threads;
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static, 16)
    for (int j = 0; j < n; ++j) {
        float out[5];
        scores[j] = -1.f;
        int lrc = 0;
        Rolled* R = (Rolled*)orc_rolled_load(paths[j], &lrc);
        int rc = pair_score(*(Latent*)lat, *R, *(Codebook*)cb, tie_mode, out);
        delete R;
        if (rc == 1) { result = 1; continue; }
        if (rc == 2) continue;
        scores[j] = out[4];
    }
    return result;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
