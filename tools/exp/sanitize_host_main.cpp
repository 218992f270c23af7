// sanitize_host_main.cpp -- driver of tools/exp/sanitize_host.sh: the threaded host code that has no HIP dependency (the ranged DataDAO reader,
// the host pool, the hub-chain schedule with its two concurrent walks) under ThreadSanitizer and AddressSanitizer + UBSan.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include "../../include/carskit_mi355x.h"
#include "../../carskit_amd/csrc/level_schedule.hpp"
#include "../../carskit_amd/csrc/host_pool.hpp"
#include <atomic>
#include <stdexcept>
int main(int argc, char **argv) {
    for (int rep = 0; rep < 3; ++rep) {
        cmi_dao_handle h = nullptr;
        int rc = cmi_dao_read(argv[1], &h);
        int64_t cnt[8];
        if (rc == 0) { cmi_dao_counts(h, cnt); printf("dao rc %d nnz %lld users %lld\n", rc, (long long)cnt[7], (long long)cnt[0]); cmi_dao_destroy(h); }
        else printf("dao rc %d %s\n", rc, cmi_dao_last_error(nullptr));
    }
    const int64_t n = 2000000; const int nu = 200000, ni = 20000;
    std::vector<int32_t> u(n), j(n);
    uint64_t s = 12345;
    for (int64_t t = 0; t < n; ++t) { s = s * 6364136223846793005ull + 1442695040888963407ull; u[t] = (s >> 33) % nu; s = s * 6364136223846793005ull + 1442695040888963407ull; j[t] = (s >> 33) % ni; }
    for (int rep = 0; rep < 2; ++rep) {
        cmi::ChainSchedule cs;
        bool ok = cmi::build_chain_schedule(n, u.data(), j.data(), nu, ni, -3, 16, cs);
        printf("chain ok %d units %lld levels %lld\n", (int)ok, (long long)cs.n_units(), (long long)cs.n_levels());
    }
    // a throwing range body: on the caller's range (0), on a worker's range, on both -- the exception must reach the caller AFTER every
    // worker is done with the std::function, and the pool must stay usable
    for (int who : {0, 3, -1}) {
        std::atomic<int> ran{0};
        bool caught = false;
        try {
            cmi::parallel_ranges(6000, 6, [&](int t, int64_t b, int64_t e) {
                std::vector<int> work((size_t)(e - b), t);
                ran++;
                if (who < 0 || t == who) throw std::runtime_error("range body failed");
            });
        } catch (const std::runtime_error &) {
            caught = true;
        }
        std::atomic<int64_t> sum{0};
        cmi::parallel_ranges(6000, 6, [&](int, int64_t b, int64_t e) { sum += e - b; });
        printf("throwing body (range %d): caught %d, ranges run %d, pool afterwards sums %lld\n", who, (int)caught, ran.load(), (long long)sum.load());
        if (!caught || sum != 6000) return 1;
    }
    return 0;
}
