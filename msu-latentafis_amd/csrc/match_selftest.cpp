// match_selftest — CPU-only checks of the host pieces of `match` (no GPU, no HIP context): used by tests/test_oracle.py and
// tests/test_sharding.py.  Kept out of the product binary.
//   match_selftest -selftest-args <opt> <tokens...>          flag parsing (cli_util.h) — pinned against matching/argparser.h
//   match_selftest -selftest-config <file> -key <key>        afis.config reader — pinned against the reference's JSON library
//   match_selftest -selftest-shards <weights file> -world N  shard cut rule (rank_exchange.cpp) — equals host/sharding.py
//   match_selftest -selftest-exchange                        rendezvous: rank 0's 128 bytes reach every rank (RANK/WORLD_SIZE/MASTER_* env)
//   match_selftest -selftest-allgather                       AFIS_EXCHANGE=tcp all-gather + the agreement point, N local ranks
//   match_selftest -selftest-groups <rows file> -per N [-plain]   the launch-group rule (afis_device.h: launch_group_cuts / launch_group_latents): cut positions for the listed latent texture row counts
//   match_selftest -selftest-classes                         the candidate kernel's shape-class rule (afis_device.h: rt_max_rows) as a table: nR  L1 L2 L4  stride1 stride2 stride4
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "afis_device.h"
#include "cli_util.h"
#include "rank_exchange.h"

using namespace afis;

int main(int argc, char** argv)
{
    ArgParser args(argc, argv);
    if (args.cmdOptionExists("-selftest-classes")) {                            // host-side class rule of k_minu_cands_rt (no GPU): tests/test_host.py checks its invariants
        std::cout << rt_class_simi_floats(1) << " " << rt_class_simi_floats(2) << " " << rt_class_simi_floats(4) << " " << rt_class_keys_per_thread(1) << " " << rt_class_keys_per_thread(2) << " "
                  << rt_class_keys_per_thread(4) << " " << rt_class_max_rolled(4) << " " << rt_class_max_latent(4) << std::endl;
        for (int nR = 0; nR <= 2000; ++nR)
            std::cout << nR << " " << rt_max_rows(1, nR) << " " << rt_max_rows(2, nR) << " " << rt_max_rows(4, nR) << " " << rt_row_stride(1, nR) << " " << rt_row_stride(2, nR) << " " << rt_row_stride(4, nR) << std::endl;
        return 0;
    }
    if (args.cmdOptionExists("-selftest-groups")) {                             // the launch-group rule (no GPU): rows per latent (one int per line), latents per launch at most
        std::ifstream f(args.getCmdOption("-selftest-groups"));
        std::vector<long long> pre{0}; long long v;
        while (f >> v) pre.push_back(pre.back() + v);
        const int n = (int)pre.size() - 1, per = atoi(args.getCmdOption("-per").c_str());
        std::vector<int> cuts((size_t)(n > 0 ? n : 1)); int m = 0;
        launch_group_cuts(pre.data(), n, per, !args.cmdOptionExists("-plain"), cuts.data(), &m);
        for (int i = 0; i < m; ++i) std::cout << cuts[(size_t)i] << (i + 1 < m ? " " : "");
        std::cout << std::endl;
        for (long long G : {1ll, 12500ll, 39000ll, 40000ll, 50000ll, 100000ll, 250000ll, 500000ll, 1000000ll}) std::cout << G << ":" << launch_group_latents(G) << " ";
        std::cout << std::endl;
        return 0;
    }
    if (args.cmdOptionExists("-selftest-exchange")) {                           // rendezvous only (no GPU): rank 0's 128 bytes reach every rank
        RankWorld w; world_from_env(w);
        unsigned char id[128];
        for (int i = 0; i < 128; ++i) id[i] = w.rank == 0 ? (unsigned char)(i * 7 + 3) : 0;
        std::string err;
        if (!tcp_broadcast(w, id, sizeof(id), err)) { std::cerr << "match: " << err << std::endl; return 2; }
        unsigned sum = 0; for (int i = 0; i < 128; ++i) sum = sum * 31 + id[i];
        std::cout << "rank " << w.rank << " of " << w.world << " id " << sum << std::endl;
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "-selftest-args")) {                      // token rules (no GPU): match -selftest-args <opt> <tokens...>
        ArgParser rest(argc - 2, argv + 2);                                     // argv[2] plays the program name, as argv[0] would
        std::cout << "exists=" << (rest.cmdOptionExists(argv[2]) ? 1 : 0) << " value=" << rest.getCmdOption(argv[2]) << std::endl;
        return 0;
    }
    if (args.cmdOptionExists("-selftest-config")) {                             // config reader (no GPU): match -selftest-config <file> -key <key>
        const auto kv = read_flat_json(args.getCmdOption("-selftest-config"));
        const auto it = kv.find(args.getCmdOption("-key"));
        std::cout << "found=" << (it != kv.end() ? 1 : 0) << " value=" << (it != kv.end() ? it->second : std::string()) << std::endl;
        return 0;
    }
    if (args.cmdOptionExists("-selftest-shards")) {                             // shard cut rule (no GPU): weights file (one int per line), world
        std::ifstream f(args.getCmdOption("-selftest-shards"));
        std::vector<int32_t> wts; int v;
        while (f >> v) wts.push_back(v);
        const int world = atoi(args.getCmdOption("-world").c_str());
        for (const auto& b : shard_bounds((int64_t)wts.size(), wts, world)) std::cout << b.first << " " << b.second << std::endl;
        std::vector<int32_t> none;
        for (const auto& b : shard_bounds((int64_t)wts.size(), none, world)) std::cout << b.first << " " << b.second << std::endl;
        return 0;
    }
    if (args.cmdOptionExists("-selftest-allgather")) {                          // tcp all-gather of rank-dependent blocks, then world_agree
        RankWorld w; world_from_env(w);
        std::string err;
        setenv("AFIS_EXCHANGE", "tcp", 1);
        if (!world_init(w, 0, err)) { std::cerr << err << std::endl; return 2; }
        const size_t n = 1000 + 17;
        std::vector<uint32_t> mine(n), all(n * (size_t)w.world);
        for (size_t i = 0; i < n; ++i) mine[i] = (uint32_t)(w.rank * 1000003u + i * 7u);
        for (int round = 0; round < 3; ++round) {
            if (!world_all_gather(w, mine.data(), all.data(), n * sizeof(uint32_t), err)) { std::cerr << err << std::endl; return 2; }
            for (int r = 0; r < w.world; ++r)
                for (size_t i = 0; i < n; ++i) if (all[(size_t)r * n + i] != (uint32_t)(r * 1000003u + i * 7u)) { std::cerr << "bad block" << std::endl; return 3; }
        }
        const int fail_rank = atoi(args.getCmdOption("-fail-rank").c_str());   // that rank reports code 7 (0: nobody fails)
        const int code = world_agree(w, (fail_rank > 0 && w.rank == fail_rank) ? 7 : 0, err);
        std::cout << "rank " << w.rank << " of " << w.world << " gathered ok, agree " << code << std::endl;
        world_finalize(w);
        return 0;
    }
    std::cerr << "match_selftest: no check named" << std::endl;
    return 2;
}
