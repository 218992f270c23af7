// level_schedule.cpp -- host-side integer preprocessing that turns the reference's sequential
// visiting order into a parallel schedule WITHOUT changing the result.
//
// The reference walks the training tuples one by one in CRS order (librec MatrixIterator; e.g.
// CAMF_CI.java:80 `for (MatrixEntry me : trainMatrix)`).  Update t reads and writes only state keyed
// by its user u_t (P[u], userBias[u], ucBias[u,*]) and by its item j_t (Q[j], itemBias[j],
// icBias[j,*]).  Two tuples with different users AND different items touch disjoint state, so they
// commute exactly.  Hence the only ordering that matters is, per user and per item, the CRS order
// of the tuples containing it.  Define
//     level(t) = 1 + max(level(previous tuple with the same user), level(previous tuple with the
//                same item)),   0 predecessors -> level 1.
// Tuples of one level are pairwise independent; running levels 1,2,3,... in sequence with all
// tuples of a level in parallel yields, for every state element, the same sequence of updates
// with the same operands as the sequential walk -- bit-identical at equal precision.
// (CAMF_C's condBias is shared by every tuple, so for CAMF_C no such schedule exists; see
// CMI_FLAG_SCHED_SERIAL.)
#include "level_schedule.hpp"

#include <system_error>
#include <cstdlib>
#include <thread>

#include "host_pool.hpp"

#include <queue>
#include <functional>
#include <utility>
#include <algorithm>
#include <numeric>

namespace cmi {

bool build_level_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items,
                          int within_level_order, LevelSchedule &out) {
    out.perm.clear();
    out.level_off.clear();
    out.max_level = 0;
    if (n <= 0) {
        out.level_off.push_back(0);
        return true;
    }
    if (n >= (int64_t)1 << 31) return false;
    std::vector<int32_t> last_u((size_t)n_users, 0), last_j((size_t)n_items, 0), level((size_t)n);
    int32_t n_levels = 0;
    for (int64_t t = 0; t < n; ++t) {
        const int32_t a = last_u[(size_t)u[t]], b = last_j[(size_t)j[t]];
        const int32_t l = (a > b ? a : b) + 1;
        last_u[(size_t)u[t]] = l;
        last_j[(size_t)j[t]] = l;
        level[(size_t)t] = l;
        if (l > n_levels) n_levels = l;
    }
    // stable counting sort by level: inside a level the CRS order is kept
    out.level_off.assign((size_t)n_levels + 1, 0);
    for (int64_t t = 0; t < n; ++t) out.level_off[(size_t)level[(size_t)t]]++;
    for (int32_t l = 1; l <= n_levels; ++l) {
        out.max_level = std::max(out.max_level, out.level_off[(size_t)l]);
        out.level_off[(size_t)l] += out.level_off[(size_t)l - 1];
    }
    out.perm.resize((size_t)n);
    {
        std::vector<int64_t> cur(out.level_off.begin(), out.level_off.end() - 1);
        for (int64_t t = 0; t < n; ++t) out.perm[(size_t)cur[(size_t)level[(size_t)t] - 1]++] = (int32_t)t;
    }
    // Tuples of a level are independent, so their order inside the level is free: sorting by item
    // (or user) id only changes which rows neighbouring lanes touch (memory locality), not the result.
    if (within_level_order == LEVEL_ORDER_XCD) {
        // XCD affinity: workgroup b of a launch lands on XCD b % 8 (observed placement, used for speed only), so give
        // workgroup b the tuples whose item id is b % 8 (mod 8): each XCD's L2 then only ever sees 1/8 of Q / icBias.
        std::vector<int32_t> bucket[8];
        for (int32_t l = 0; l < n_levels; ++l) {
            const int64_t b = out.level_off[(size_t)l], e = out.level_off[(size_t)l + 1];
            for (auto &v : bucket) v.clear();
            for (int64_t q = b; q < e; ++q) bucket[j[out.perm[(size_t)q]] & 7].push_back(out.perm[(size_t)q]);
            size_t taken[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int64_t w = b;
            for (int64_t blk = 0; w < e; ++blk) {
                int x = (int)(blk & 7), tries = 0;
                while (taken[x] >= bucket[x].size() && tries < 8) { // bucket exhausted: borrow from the next one
                    x = (x + 1) & 7;
                    ++tries;
                }
                for (int c = 0; c < LEVEL_XCD_BLOCK && w < e; ++c) {
                    while (taken[x] >= bucket[x].size()) x = (x + 1) & 7;
                    out.perm[(size_t)w++] = bucket[x][taken[x]++];
                }
            }
        }
    } else if (within_level_order != LEVEL_ORDER_CRS) {
        const int32_t *key = within_level_order == LEVEL_ORDER_ITEM ? j : u;
        for (int32_t l = 0; l < n_levels; ++l) {
            auto b = out.perm.begin() + out.level_off[(size_t)l], e = out.perm.begin() + out.level_off[(size_t)l + 1];
            std::sort(b, e, [key](int32_t x, int32_t y) { return key[x] != key[y] ? key[x] < key[y] : x < y; });
        }
    }
    return true;
}

int64_t build_narrow_runs(const std::vector<int64_t> &level_off, int64_t max_tuples, int64_t min_levels, std::vector<int32_t> &run_len) {
    const int64_t n_levels = (int64_t)level_off.size() - 1;
    run_len.assign((size_t)(n_levels > 0 ? n_levels : 0), 0);
    int64_t launches = 0;
    for (int64_t l = 0; l < n_levels;) {
        int64_t e = l;
        while (e < n_levels && e - l < ((int64_t)1 << 30) && level_off[(size_t)e + 1] - level_off[(size_t)e] <= max_tuples) ++e;
        if (e - l >= min_levels) {
            run_len[(size_t)l] = (int32_t)(e - l);
            for (int64_t q = l + 1; q < e; ++q) run_len[(size_t)q] = -1;
            ++launches;
            l = e;
        } else {
            const int64_t stop = e > l ? e : l + 1; // these levels keep their own launches
            launches += stop - l;
            l = stop;
        }
    }
    return launches;
}

void build_conflict_free_blocks(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int max_block,
                                std::vector<int32_t> &off) {
    std::vector<int32_t> seen_u((size_t)n_users, -1), seen_j((size_t)n_items, -1);
    off.assign(1, 0);
    int32_t cur = 0, len = 0;
    for (int64_t t = 0; t < n; ++t) {
        if (len == max_block || seen_u[(size_t)u[t]] == cur || seen_j[(size_t)j[t]] == cur) {
            off.push_back((int32_t)t);
            ++cur;
            len = 0;
        }
        seen_u[(size_t)u[t]] = cur;
        seen_j[(size_t)j[t]] = cur;
        ++len;
    }
    if (n > 0) off.push_back((int32_t)n);
}

// ---------------------------------------------------------------------------------------------------------------------
// Hub-chain level schedule.
//
// The plain level schedule puts two tuples of the same item (or user) into different levels, so a row that is updated m
// times per epoch is read and written m times from HBM and the epoch has at least m dependent launches.  But the reference's
// order only says that the tuples of one row run in CRS order -- it does not say that somebody else has to run in between.
// So let consecutive tuples of one HUB row (say item j: tuples t1 < t2 < ... in CRS order) be executed back to back by the
// same 16-lane group, with Q[j], itemBias[j] and icBias[j,:] kept on chip between them, whenever the OTHER row of the later
// tuple (its user) is already final, i.e. the user's previous tuple sits in a strictly earlier level:
//     A = level of the hub row's previous tuple, B = level of the spoke row's previous tuple
//     A > B and the hub's current unit has room  ->  the tuple joins that unit (same level A)
//     otherwise                                  ->  it opens a new unit in level max(A, B) + 1.
// Invariants (property-tested in tests/test_level_schedule.py): two tuples of one level that share a user or an item are in
// the same unit; a unit's tuples share the hub, are consecutive in the hub's CRS chain and have pairwise distinct spokes
// (equal spokes would give A == B); for every user and every item the (level, position in unit) order of its tuples is the
// CRS order.  Hence executing levels in sequence, units of a level in parallel, and a unit's tuples in order applies to
// every state element the same updates with the same operands as the sequential walk -- the same result as the plain
// level schedule, bit for bit at equal arithmetic.
// Inside a level the units are sorted by length, longest first (free: they are independent): the 4 groups of a wave then
// walk chains of similar length and the longest chains are dispatched first.
// The recurrence is sequential in t, but what it costs is the cache miss on the rows' level entries (a 10 M-user table is 40 MB): the
// entries of tuple t + 24 are requested while tuple t is processed, and a hub row's three fields share one 12-byte entry.
namespace {
struct HubEntry {
    int32_t level; // level of the hub row's previous tuple
    int32_t unit;  // its current unit
    int32_t len;   // tuples in that unit so far
};
} // namespace
static void chain_pass(int64_t n, const int32_t *hub, const int32_t *spoke, int32_t n_hub, int32_t n_spoke, int max_chain,
                       std::vector<int32_t> *unit_out, std::vector<uint8_t> *pos_out, std::vector<int32_t> *unit_level,
                       std::vector<uint8_t> *unit_len, int32_t &n_levels, int64_t &n_units) {
    std::vector<HubEntry> hb((size_t)n_hub, HubEntry{0, -1, 0});
    std::vector<int32_t> ls((size_t)n_spoke, 0);
    n_levels = 0;
    n_units = 0;
    constexpr int64_t AHEAD = 24; // (8 / 24 / 64 / 128 measured on an MI355X box's host, 100 M tuples: 2.4 / 2.0 / 2.0 / 2.1 s)
    for (int64_t t = 0; t < n; ++t) {
        if (t + AHEAD < n) {
            __builtin_prefetch(&hb[(size_t)hub[t + AHEAD]], 1);
            __builtin_prefetch(&ls[(size_t)spoke[t + AHEAD]], 1);
        }
        HubEntry &h = hb[(size_t)hub[t]];
        const size_t ss = (size_t)spoke[t];
        const int32_t A = h.level, B = ls[ss];
        int32_t l;
        if (A > B && h.len < max_chain) {
            l = A;
            if (pos_out) (*pos_out)[(size_t)t] = (uint8_t)h.len;
            h.len++;
            if (unit_len) (*unit_len)[(size_t)h.unit]++;
        } else {
            l = (A > B ? A : B) + 1;
            h.len = 1;
            h.unit = (int32_t)n_units++;
            if (pos_out) (*pos_out)[(size_t)t] = 0;
            if (unit_level) {
                unit_level->push_back(l);
                unit_len->push_back(1);
            }
        }
        h.level = ls[ss] = l;
        if (unit_out) (*unit_out)[(size_t)t] = h.unit;
        if (l > n_levels) n_levels = l;
    }
}

// ---------------------------------------------------------------------------------------------
// owner (dataflow) schedule
// ---------------------------------------------------------------------------------------------
// Heavy-tailed degrees make the dependency levels narrow: the hottest row's tuples form one chain as long as its degree, and a
// level-synchronous epoch pays a launch (or a workgroup barrier) per link.  Here every hub row has ONE owner for the whole epoch.  An
// owner walks the tuples of all its rows in CRS order, so
//   * the chain along a hub row never leaves the owner's registers (OWN_HUB_FWD: same hub row as the owner's previous tuple);
//   * the only state that crosses owners is the spoke row, whose record in HBM carries, in every 8-byte granule, the number of
//     updates applied to it so far this epoch; a tuple may use the record once every granule carries want[pos];
//   * the tuple with the smallest CRS index that has not run yet is always at the head of its owner's list and all its predecessors
//     have run, so with every owner resident the epoch cannot deadlock.
// Rows are dealt to owners longest-processing-time-first (a heap over the owners' loads), so the hottest rows sit alone.
bool build_owner_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub, int n_owners,
                          int depth, OwnerSchedule &out) {
    if (n < 0 || n >= ((int64_t)1 << 31) - 1024 || n_owners < 1 || depth < 1) return false;
    std::vector<int64_t> deg_u((size_t)n_users, 0), deg_j((size_t)n_items, 0);
    for (int64_t t = 0; t < n; ++t) {
        deg_u[(size_t)u[t]]++;
        deg_j[(size_t)j[t]]++;
    }
    if (hub < 0) {
        int64_t mu = 0, mj = 0;
        for (int64_t d : deg_u) mu = std::max(mu, d);
        for (int64_t d : deg_j) mj = std::max(mj, d);
        hub = mj >= mu ? 1 : 0;
    }
    out.hub_is_item = hub ? 1 : 0;
    const int32_t *hv = hub ? j : u, *sv = hub ? u : j;
    const int32_t n_hubs = hub ? n_items : n_users, n_spokes = hub ? n_users : n_items;
    const std::vector<int64_t> &deg = hub ? deg_j : deg_u;

    // longest-processing-time-first: rows by degree (descending, ties by id), each to the least loaded owner (ties by owner id)
    std::vector<int32_t> order;
    order.reserve((size_t)n_hubs);
    for (int32_t x = 0; x < n_hubs; ++x)
        if (deg[(size_t)x] > 0) order.push_back(x);
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return deg[(size_t)a] != deg[(size_t)b] ? deg[(size_t)a] > deg[(size_t)b] : a < b; });
    typedef std::pair<int64_t, int32_t> Load; // (tuples, owner)
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int32_t w = 0; w < n_owners; ++w) heap.push(Load(0, w));
    std::vector<int32_t> owner((size_t)n_hubs, -1);
    std::vector<int64_t> load((size_t)n_owners, 0);
    for (int32_t x : order) {
        Load l = heap.top();
        heap.pop();
        owner[(size_t)x] = l.second;
        l.first += deg[(size_t)x];
        load[(size_t)l.second] = l.first;
        heap.push(l);
    }
    out.own_off.assign((size_t)n_owners + 1, 0);
    out.max_load = 0;
    for (int32_t w = 0; w < n_owners; ++w) {
        out.own_off[(size_t)w + 1] = out.own_off[(size_t)w] + load[(size_t)w];
        out.max_load = std::max(out.max_load, load[(size_t)w]);
    }
    // the owners' lists, each in CRS order; want = the spoke row's update count before the tuple
    out.perm.assign((size_t)n, 0);
    out.want.assign((size_t)n, 0);
    out.flags.assign((size_t)n, 0);
    {
        std::vector<int64_t> cur(out.own_off.begin(), out.own_off.end() - 1);
        std::vector<uint32_t> seen((size_t)n_spokes, 0);
        for (int64_t t = 0; t < n; ++t) {
            const int64_t pos = cur[(size_t)owner[(size_t)hv[t]]]++;
            out.perm[(size_t)pos] = (int32_t)t;
            out.want[(size_t)pos] = seen[(size_t)sv[t]]++;
        }
    }
    std::vector<int64_t> last_pos((size_t)n_hubs, -1); // list position of the hub row's previous tuple
    for (int32_t w = 0; w < n_owners; ++w) {
        const int64_t b = out.own_off[(size_t)w], e = out.own_off[(size_t)w + 1];
        for (int64_t pos = b; pos < e; ++pos) {
            const int64_t t = out.perm[(size_t)pos];
            uint32_t f = 0;
            const int64_t lp = last_pos[(size_t)hv[t]];
            if (lp >= 0) {
                if (pos - lp == 1) f |= OWN_HUB_FWD;
                else if (pos - lp <= depth) f |= OWN_HUB_LATE;
            }
            last_pos[(size_t)hv[t]] = pos;
            if (pos > b) {
                const int64_t tp = out.perm[(size_t)pos - 1];
                if (sv[tp] == sv[t] && out.want[(size_t)pos] == out.want[(size_t)pos - 1] + 1) f |= OWN_SPK_FWD;
            }
            out.flags[(size_t)pos] = f;
        }
        for (int64_t pos = b; pos < e; ++pos) { // a row goes back to HBM when the next tuple does not take it over in registers
            const uint32_t nf = pos + 1 < e ? out.flags[(size_t)pos + 1] : 0u;
            if (!(nf & OWN_HUB_FWD)) out.flags[(size_t)pos] |= OWN_HUB_STORE;
            if (!(nf & OWN_SPK_FWD)) out.flags[(size_t)pos] |= OWN_SPK_STORE;
        }
    }
    return true;
}


int64_t count_plain_levels(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items) {
    if (n <= 0) return 0;
    int32_t nl = 0;
    int64_t nu = 0;
    chain_pass(n, j, u, n_items, n_users, 1, nullptr, nullptr, nullptr, nullptr, nl, nu); // max_chain 1 = the plain level recurrence
    return nl;
}

bool build_chain_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub, int max_chain,
                          ChainSchedule &out) {
    out = ChainSchedule();
    out.unit_off.push_back(0);
    out.level_off.push_back(0);
    if (max_chain < 1) max_chain = 1;
    if (max_chain > 255) max_chain = 255;
    out.hub_is_item = hub == 0 ? 0 : 1;
    if (n <= 0) return true;
    if (n >= (int64_t)1 << 31) return false;
    struct Side {
        std::vector<int32_t> unit_of, unit_level;
        std::vector<uint8_t> pos, unit_len;
        int32_t nl = 0;
        int64_t nu = 0;
    };
    auto run_side = [&](int item_hub, Side &sd) {
        sd.unit_of.resize((size_t)n);
        sd.pos.resize((size_t)n);
        if (item_hub) chain_pass(n, j, u, n_items, n_users, max_chain, &sd.unit_of, &sd.pos, &sd.unit_level, &sd.unit_len, sd.nl, sd.nu);
        else chain_pass(n, u, j, n_users, n_items, max_chain, &sd.unit_of, &sd.pos, &sd.unit_level, &sd.unit_len, sd.nl, sd.nu);
    };
    Side side;
    auto pick = [&](int64_t units_item, int64_t units_user) {
        if (hub == -2) return (double)units_user <= 1.3 * (double)units_item ? 0 : 1;
        if (hub == -3) return (double)units_item <= 1.3 * (double)units_user ? 1 : 0;
        return units_item <= units_user ? 1 : 0;
    };
    if (hub < 0 && n > ((int64_t)32 << 20)) {
        // Large sets: the side is chosen on the FIRST EIGHTH of the tuples (both sides counted there, side by side), then only that
        // side is walked over all of them -- the two full walks ran at twice the time of one (they share the memory system), and the
        // choice moves the epoch's time, never its result (either side's schedule is order-exact).
        const int64_t m = n / 8;
        int32_t l1 = 0, l0 = 0;
        int64_t ui = 0, uu = 0;
        bool threaded = true;
        std::thread th;
        try {
            th = std::thread([&]() { chain_pass(m, u, j, n_users, n_items, max_chain, nullptr, nullptr, nullptr, nullptr, l0, uu); });
        } catch (const std::system_error &) {
            threaded = false;
        }
        chain_pass(m, j, u, n_items, n_users, max_chain, nullptr, nullptr, nullptr, nullptr, l1, ui);
        if (threaded) th.join();
        else chain_pass(m, u, j, n_users, n_items, max_chain, nullptr, nullptr, nullptr, nullptr, l0, uu);
        hub = pick(ui, uu);
    }
    if (hub < 0) { // fewer units = fewer hub-row round trips through HBM; -2 / -3: a preferred side wins up to 1.3x the other's units
        // both sides are walked at the same time, each with its full output; the loser's is dropped (before: two counting walks, then
        // the winner's walk again -- three sequential passes over the tuples)
        Side other;
        bool threaded = true;
        std::thread th;
        try {
            th = std::thread([&]() { run_side(0, other); });
        } catch (const std::system_error &) { // the process may not create more threads: one side after the other
            threaded = false;
        }
        run_side(1, side);
        if (threaded) th.join();
        else run_side(0, other);
        hub = pick(side.nu, other.nu);
        if (!hub) std::swap(side, other);
    } else run_side(hub ? 1 : 0, side);
    out.hub_is_item = hub ? 1 : 0;
    const int32_t nl = side.nl;
    const int64_t nu = side.nu;
    const std::vector<int32_t> &unit_of = side.unit_of, &unit_level = side.unit_level;
    const std::vector<uint8_t> &unit_len = side.unit_len, &pos = side.pos;
    // counting sort of the units by (level ascending, length descending), stable in unit id (= CRS order of the first tuple)
    const size_t nkeys = (size_t)nl * (size_t)max_chain;
    std::vector<int64_t> key_off(nkeys + 1, 0);
    auto key = [&](int64_t q) { return (size_t)(unit_level[(size_t)q] - 1) * (size_t)max_chain + (size_t)(max_chain - unit_len[(size_t)q]); };
    for (int64_t q = 0; q < nu; ++q) key_off[key(q) + 1]++;
    for (size_t x = 0; x < nkeys; ++x) key_off[x + 1] += key_off[x];
    out.level_off.assign((size_t)nl + 1, 0);
    for (int32_t l = 0; l <= nl; ++l) out.level_off[(size_t)l] = key_off[(size_t)l * (size_t)max_chain];
    for (int32_t l = 0; l < nl; ++l) out.max_level_units = std::max(out.max_level_units, out.level_off[(size_t)l + 1] - out.level_off[(size_t)l]);
    std::vector<int32_t> rank((size_t)nu); // unit id -> position in the schedule
    {
        std::vector<int64_t> cur(key_off.begin(), key_off.end() - 1);
        for (int64_t q = 0; q < nu; ++q) rank[(size_t)q] = (int32_t)cur[key(q)]++;
    }
    out.unit_off.assign((size_t)nu + 1, 0);
    for (int64_t q = 0; q < nu; ++q) out.unit_off[(size_t)rank[(size_t)q] + 1] = unit_len[(size_t)q];
    for (int64_t q = 0; q < nu; ++q) out.unit_off[(size_t)q + 1] += out.unit_off[(size_t)q];
    // a tuple's place: its unit's first slot + its position inside the unit (recorded by the walk) -- independent per tuple
    out.perm.resize((size_t)n);
    parallel_ranges(n, host_threads(n), [&](int, int64_t b, int64_t e) {
        for (int64_t t = b; t < e; ++t)
            out.perm[(size_t)out.unit_off[(size_t)rank[(size_t)unit_of[(size_t)t]]] + pos[(size_t)t]] = (int32_t)t;
    });
    return true;
}

} // namespace cmi
