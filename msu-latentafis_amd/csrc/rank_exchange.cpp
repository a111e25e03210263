// rank_exchange.cpp — see rank_exchange.h.  Host code: HIP runtime for the staging buffers, RCCL for the one collective.
#include "rank_exchange.h"

#include <arpa/inet.h>
#include <hip/hip_runtime.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <rccl/rccl.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <thread>

namespace afis {

static int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

void world_from_env(RankWorld& w)
{
    w.rank = env_int("RANK", 0);
    w.world = std::max(1, env_int("WORLD_SIZE", 1));
    w.local_rank = env_int("LOCAL_RANK", w.rank);
    const char* a = getenv("MASTER_ADDR");
    if (a && *a) w.addr = a;
    // MASTER_PORT itself belongs to the launcher's own store; a launcher that has verified a free port for this rendezvous passes it as AFIS_EXCHANGE_PORT
    w.port = env_int("AFIS_EXCHANGE_PORT", env_int("MASTER_PORT", 29500) + 1);
    if (w.rank < 0 || w.rank >= w.world) { w.rank = 0; w.world = 1; }
}

static bool send_all(int fd, const void* p, size_t n)
{
    const char* b = (const char*)p;
    while (n) { const ssize_t k = ::send(fd, b, n, MSG_NOSIGNAL); if (k <= 0) return false; b += k; n -= (size_t)k; }
    return true;
}
static bool recv_all(int fd, void* p, size_t n)
{
    char* b = (char*)p;
    while (n) { const ssize_t k = ::recv(fd, b, n, 0); if (k <= 0) return false; b += k; n -= (size_t)k; }
    return true;
}

bool tcp_broadcast(const RankWorld& w, void* buf, size_t len, std::string& err)
{
    if (w.world <= 1) return true;
    sockaddr_in sa; memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)w.port);
    if (w.rank == 0) {
        const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) { err = "socket() failed"; return false; }
        const int one = 1; setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sa.sin_addr.s_addr = htonl(INADDR_ANY);
        if (::bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0 || ::listen(ls, w.world) != 0) { ::close(ls); err = "cannot listen on port " + std::to_string(w.port); return false; }
        bool ok = true;
        for (int i = 1; i < w.world && ok; ++i) {
            const int c = ::accept(ls, nullptr, nullptr);
            if (c < 0) { ok = false; break; }
            int32_t peer = -1;
            ok = recv_all(c, &peer, sizeof(peer)) && peer > 0 && peer < w.world && send_all(c, buf, len);
            ::close(c);
        }
        ::close(ls);
        if (!ok) err = "rank 0: id exchange with a peer failed";
        return ok;
    }
    if (inet_pton(AF_INET, w.addr.c_str(), &sa.sin_addr) != 1) { err = "MASTER_ADDR must be an IPv4 address, got " + w.addr; return false; }
    for (int attempt = 0; attempt < 600; ++attempt) {                       // rank 0 may still be starting: retry for ~60 s
        const int c = ::socket(AF_INET, SOCK_STREAM, 0);
        if (c < 0) { err = "socket() failed"; return false; }
        if (::connect(c, (sockaddr*)&sa, sizeof(sa)) == 0) {
            const int32_t me = w.rank;
            const bool ok = send_all(c, &me, sizeof(me)) && recv_all(c, buf, len);
            ::close(c);
            if (!ok) err = "rank " + std::to_string(w.rank) + ": id exchange with rank 0 failed";
            return ok;
        }
        ::close(c);
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    err = "rank " + std::to_string(w.rank) + ": cannot reach rank 0 at " + w.addr + ":" + std::to_string(w.port);
    return false;
}

#define RX_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return false; } } while (0)
#define RX_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { err = std::string(#call) + ": " + ncclGetErrorString(r_); return false; } } while (0)

static void set_timeouts(int fd, double seconds)
{
    if (seconds < 0.001) seconds = 0.001;                                  // {0,0} would mean "no timeout"
    timeval tv; tv.tv_sec = (long)seconds; tv.tv_usec = (long)((seconds - (double)tv.tv_sec) * 1e6);
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}

// AFIS_EXCHANGE=tcp: every rank's block goes to rank 0, the concatenation comes back.  Header per peer: rank, call counter, size.
static bool tcp_all_gather(RankWorld& w, const void* send, void* recv, size_t bytes, std::string& err)
{
    struct Hdr { int32_t rank; uint32_t seq; uint64_t bytes; };
    const uint32_t seq = w.seq++;
    if (w.rank == 0) {
        if (w.listen_fd < 0) {
            sockaddr_in sa; memset(&sa, 0, sizeof(sa));
            sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)w.port); sa.sin_addr.s_addr = htonl(INADDR_ANY);
            const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
            if (ls < 0) { err = "socket() failed"; return false; }
            const int one = 1; setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
            if (::bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0 || ::listen(ls, 64) != 0) { ::close(ls); err = "cannot listen on port " + std::to_string(w.port); return false; }
            w.listen_fd = ls;
        }
        memcpy(recv, send, bytes);
        std::vector<int> peers;
        std::vector<char> seen((size_t)w.world, 0);                         // every rank contributes exactly one block per exchange
        bool ok = true;
        for (int i = 1; i < w.world && ok; ++i) {
            pollfd pf{w.listen_fd, POLLIN, 0};
            if (::poll(&pf, 1, (int)(w.timeout_s * 1000)) <= 0) { err = "rank 0: timed out waiting for a peer in exchange " + std::to_string(seq); ok = false; break; }
            const int c = ::accept(w.listen_fd, nullptr, nullptr);
            if (c < 0) { err = "accept() failed"; ok = false; break; }
            set_timeouts(c, w.timeout_s);
            peers.push_back(c);
            Hdr h{};
            ok = recv_all(c, &h, sizeof(h)) && h.rank > 0 && h.rank < w.world && !seen[(size_t)h.rank] && h.seq == seq && h.bytes == bytes &&
                 recv_all(c, (char*)recv + (size_t)h.rank * bytes, bytes);
            if (ok) seen[(size_t)h.rank] = 1;
            else err = "rank 0: a peer sent a mismatching or duplicate block in exchange " + std::to_string(seq);
        }
        for (int c : peers) { if (ok && !send_all(c, recv, bytes * w.world)) { ok = false; err = "rank 0: returning the gathered blocks failed"; } ::close(c); }
        return ok;
    }
    sockaddr_in sa; memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)w.port);
    if (inet_pton(AF_INET, w.addr.c_str(), &sa.sin_addr) != 1) { err = "MASTER_ADDR must be an IPv4 address, got " + w.addr; return false; }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(w.timeout_s);
    while (std::chrono::steady_clock::now() < deadline) {
        const int c = ::socket(AF_INET, SOCK_STREAM, 0);
        if (c < 0) { err = "socket() failed"; return false; }
        if (::connect(c, (sockaddr*)&sa, sizeof(sa)) == 0) {
            set_timeouts(c, w.timeout_s);
            const Hdr h{w.rank, seq, (uint64_t)bytes};
            const bool ok = send_all(c, &h, sizeof(h)) && send_all(c, send, bytes) && recv_all(c, recv, bytes * w.world);
            ::close(c);
            if (!ok) err = "rank " + std::to_string(w.rank) + ": exchange " + std::to_string(seq) + " with rank 0 failed (peer gone or timed out)";
            return ok;
        }
        ::close(c);
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    err = "rank " + std::to_string(w.rank) + ": cannot reach rank 0 at " + w.addr + ":" + std::to_string(w.port);
    return false;
}

bool world_init(RankWorld& w, int device, std::string& err)
{
    if (const char* t = getenv("AFIS_EXCHANGE_TIMEOUT_S")) { const double v = atof(t); if (v > 0) w.timeout_s = v; }
    const char* ex = getenv("AFIS_EXCHANGE");
    w.tcp = ex && !strcmp(ex, "tcp");
    if (w.tcp) return true;
    RX_HIP(hipSetDevice(device));
    ncclUniqueId id; memset(&id, 0, sizeof(id));
    if (w.rank == 0) RX_NCCL(ncclGetUniqueId(&id));
    if (!tcp_broadcast(w, &id, sizeof(id), err)) return false;
    ncclComm_t comm = nullptr;
    RX_NCCL(ncclCommInitRank(&comm, w.world, id, w.rank));
    w.comm = comm;
    hipStream_t s = nullptr;
    RX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    w.stream = s;
    return true;
}

bool world_all_gather(RankWorld& w, const void* send, void* recv, size_t bytes, std::string& err)
{
    if (bytes == 0) return true;
    if (w.tcp) return tcp_all_gather(w, send, recv, bytes, err);
    if (!w.comm) { err = "world_all_gather: communicator not initialised"; return false; }
    hipStream_t s = (hipStream_t)w.stream;
    if (w.cap_send < bytes) { if (w.d_send) RX_HIP(hipFree(w.d_send)); w.d_send = nullptr; RX_HIP(hipMalloc(&w.d_send, bytes)); w.cap_send = bytes; }
    if (w.cap_recv < bytes * w.world) { if (w.d_recv) RX_HIP(hipFree(w.d_recv)); w.d_recv = nullptr; RX_HIP(hipMalloc(&w.d_recv, bytes * w.world)); w.cap_recv = bytes * w.world; }
    RX_HIP(hipMemcpyAsync(w.d_send, send, bytes, hipMemcpyHostToDevice, s));
    RX_NCCL(ncclAllGather(w.d_send, w.d_recv, bytes, ncclChar, (ncclComm_t)w.comm, s));
    RX_HIP(hipMemcpyAsync(recv, w.d_recv, bytes * w.world, hipMemcpyDeviceToHost, s));
    // bounded wait: a peer that died before the collective would otherwise leave this rank in it for ever
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(w.timeout_s);
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return true;
        if (q != hipErrorNotReady) { err = std::string("exchange failed: ") + hipGetErrorString(q); (void)hipStreamSynchronize(s); return false; }
        if (std::chrono::steady_clock::now() > deadline) {
            err = "exchange timed out after " + std::to_string((int)w.timeout_s) + " s (a peer rank is gone?)";
            (void)ncclCommAbort((ncclComm_t)w.comm); w.comm = nullptr;
            (void)hipStreamSynchronize(s);                                 // the D2H copy into the caller's buffer may still be queued: nothing may touch `recv` after we return
            return false;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

int world_agree(RankWorld& w, int my_code, std::string& err)
{
    if (w.world <= 1 && !w.tcp && !w.comm) return my_code;
    std::vector<int32_t> all((size_t)w.world, 0);
    const int32_t mine = my_code;
    if (!world_all_gather(w, &mine, all.data(), sizeof(mine), err)) return -1000;
    for (int32_t c : all) if (c != 0) return c;
    return 0;
}

void world_finalize(RankWorld& w)
{
    if (w.d_send) (void)hipFree(w.d_send);
    if (w.d_recv) (void)hipFree(w.d_recv);
    if (w.comm) (void)ncclCommDestroy((ncclComm_t)w.comm);
    if (w.stream) (void)hipStreamDestroy((hipStream_t)w.stream);
    if (w.listen_fd >= 0) ::close(w.listen_fd);
    w.listen_fd = -1;
    w.d_send = w.d_recv = w.comm = w.stream = nullptr; w.cap_send = w.cap_recv = 0;
}

std::vector<std::pair<int64_t, int64_t>> shard_bounds(int64_t G, const std::vector<int32_t>& weights, int world)
{
    std::vector<std::pair<int64_t, int64_t>> out;
    if (world <= 1) { out.emplace_back(0, G); return out; }
    std::vector<double> c((size_t)G + 1, 0.0);                              // cumulative cost; unit cost when no weights are given
    for (int64_t i = 0; i < G; ++i) c[(size_t)i + 1] = c[(size_t)i] + (weights.empty() ? 1.0 : (double)weights[(size_t)i]);
    std::vector<int64_t> cuts{0};
    for (int r = 1; r < world; ++r) {
        const double target = c[(size_t)G] * r / world;
        int64_t k = std::lower_bound(c.begin(), c.end(), target) - c.begin();   // numpy searchsorted(side="left")
        k = std::min(std::max(k, cuts.back()), G);
        cuts.push_back(k);
    }
    cuts.push_back(G);
    for (int r = 0; r < world; ++r) out.emplace_back(cuts[(size_t)r], cuts[(size_t)r + 1]);
    return out;
}

void merge_topk(const std::vector<int64_t>& idx, const std::vector<float>& score, int world, int k, int k_out,
                std::vector<int64_t>& out_idx, std::vector<float>& out_score)
{
    std::vector<size_t> ord;
    for (size_t i = 0; i < (size_t)world * k; ++i) if (idx[i] >= 0) ord.push_back(i);
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return score[a] > score[b] || (score[a] == score[b] && idx[a] < idx[b]); });
    out_idx.assign((size_t)k_out, -1); out_score.assign((size_t)k_out, -INFINITY);
    for (size_t r = 0; r < ord.size() && r < (size_t)k_out; ++r) { out_idx[r] = idx[ord[r]]; out_score[r] = score[ord[r]]; }
}

}  // namespace afis

// ---- C entry points (libafis_exchange.so): the SAME exchange step for hosts that are not the C++ `match` binary — bench.py --exchange cpp
// binds them with ctypes so that the timed step contains this file's ncclAllGather, not torch.distributed's.
extern "C" {

struct afis_exchange { afis::RankWorld w; std::string err; };

// ranks from the environment (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT: the rendezvous uses MASTER_PORT + 1); NULL on failure
afis_exchange* afis_exchange_create(int device, char* errbuf, size_t errcap)
{
    afis_exchange* x = new afis_exchange();
    afis::world_from_env(x->w);
    if (!afis::world_init(x->w, device, x->err)) {
        if (errbuf && errcap) { strncpy(errbuf, x->err.c_str(), errcap - 1); errbuf[errcap - 1] = 0; }
        afis::world_finalize(x->w); delete x; return nullptr;
    }
    return x;
}
int afis_exchange_world(const afis_exchange* x) { return x ? x->w.world : 0; }
int afis_exchange_rank(const afis_exchange* x) { return x ? x->w.rank : -1; }
// 1 = RCCL (ncclAllGather), 0 = the TCP stand-in (AFIS_EXCHANGE=tcp)
int afis_exchange_is_rccl(const afis_exchange* x) { return x && !x->w.tcp ? 1 : 0; }
// every rank contributes `bytes` bytes; recv gets world * bytes, rank-major.  0 = ok
int afis_exchange_all_gather(afis_exchange* x, const void* send, void* recv, size_t bytes)
{
    if (!x || !send || !recv) return -1;
    return afis::world_all_gather(x->w, send, recv, bytes, x->err) ? 0 : -2;
}
// what RCCL itself says about the communicator: ncclCommCount (ranks), ncclCommCuDevice (the HIP device this rank's communicator is bound to);
// -1 with the TCP stand-in (no communicator) or on error.  bench.py puts both into its JSON line so that a multi-GPU run certifies itself.
int afis_exchange_comm_count(const afis_exchange* x)
{
    int n = -1;
    if (!x || x->w.tcp || !x->w.comm) return -1;
    return ncclCommCount((ncclComm_t)x->w.comm, &n) == ncclSuccess ? n : -1;
}
int afis_exchange_comm_device(const afis_exchange* x)
{
    int d = -1;
    if (!x || x->w.tcp || !x->w.comm) return -1;
    return ncclCommCuDevice((ncclComm_t)x->w.comm, &d) == ncclSuccess ? d : -1;
}
const char* afis_exchange_last_error(const afis_exchange* x) { return x ? x->err.c_str() : "null handle"; }
void afis_exchange_destroy(afis_exchange* x) { if (x) { afis::world_finalize(x->w); delete x; } }

}  // extern "C"
