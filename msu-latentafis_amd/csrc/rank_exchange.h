// rank_exchange.h — the multi-GPU side of the `match` host: one process per GPU, gallery sharded by contiguous
// template ranges, ONE exchange step per query batch — an RCCL all-gather (xGMI inside a node) of fixed-size per-rank blocks:
// the per-shard top-24 rank lists in -l mode, the per-shard score columns in -ldir mode.  No other collective exists in the path.
// Ranks come from the environment every launcher sets (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT), e.g.
//   python -m torch.distributed.run --no-python --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 ./match ...
// The 128-byte ncclUniqueId travels from rank 0 to the others over one TCP connection each (MASTER_ADDR : MASTER_PORT + 1).
// AFIS_EXCHANGE=tcp replaces the collective by a gather-and-return through rank 0 over the same TCP port: no RCCL, so several ranks
// can share one GPU — the N > 1 code path of `match` on a 1-GPU test box (the blocks are 29 KB .. a few MB).  RCCL is the
// production path.  Either way an exchange that does not complete within AFIS_EXCHANGE_TIMEOUT_S seconds (default 600) fails
// instead of blocking for ever (a peer that died would otherwise leave the others in the collective).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace afis {

struct RankWorld {
    int rank = 0, world = 1, local_rank = 0;
    std::string addr = "127.0.0.1";
    int port = 29501;
    void* comm = nullptr;      // ncclComm_t
    void* stream = nullptr;    // hipStream_t
    void* d_send = nullptr; void* d_recv = nullptr; size_t cap_send = 0, cap_recv = 0;
    bool tcp = false;          // AFIS_EXCHANGE=tcp
    int listen_fd = -1;        // rank 0, tcp mode: kept open between exchanges
    uint32_t seq = 0;          // exchange counter (tcp mode: every rank must be in the same call)
    double timeout_s = 600.0;
};

void world_from_env(RankWorld& w);                                           // defaults: a single rank
// rank 0 -> every other rank: `len` bytes (the ncclUniqueId); plain TCP, blocking, with connect retries for ~60 s
bool tcp_broadcast(const RankWorld& w, void* buf, size_t len, std::string& err);
bool world_init(RankWorld& w, int device, std::string& err);                 // id exchange + ncclCommInitRank on `device`
// every rank contributes `bytes` bytes; recv (host) gets world * bytes, rank-major.  Staged through device buffers: the
// collective itself is ncclAllGather on the communicator's stream.
bool world_all_gather(RankWorld& w, const void* send, void* recv, size_t bytes, std::string& err);
// every rank passes its own status code; returns the first non-zero code in rank order (0 if all are zero), or -1000 if the
// exchange itself failed.  A rank that failed locally still takes part, so the whole job stops together instead of hanging.
int world_agree(RankWorld& w, int my_code, std::string& err);
void world_finalize(RankWorld& w);

// contiguous shards [lo, hi) over G templates, balanced by `weights` (per-template texture point counts) or, when weights is
// empty, by template count — the same cut rule as msu-latentafis_amd/host/sharding.py::shard_bounds
std::vector<std::pair<int64_t, int64_t>> shard_bounds(int64_t G, const std::vector<int32_t>& weights, int world);

// merge of per-rank top-k lists (rank-major [world][k]; idx < 0 = padding): score descending, index ascending
void merge_topk(const std::vector<int64_t>& idx, const std::vector<float>& score, int world, int k, int k_out,
                std::vector<int64_t>& out_idx, std::vector<float>& out_score);

}  // namespace afis
