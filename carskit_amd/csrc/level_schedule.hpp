// level_schedule.hpp -- see level_schedule.cpp
#pragma once
#include <stdint.h>
#include <vector>

namespace cmi {

enum { LEVEL_ORDER_CRS = 0, LEVEL_ORDER_ITEM = 1, LEVEL_ORDER_USER = 2, LEVEL_ORDER_XCD = 3 };
static const int LEVEL_XCD_BLOCK = 32; // tuples per workgroup of the fast level kernel (16 groups x 2 tuples)

struct LevelSchedule {
    std::vector<int32_t> perm;       // schedule position -> CRS tuple index
    std::vector<int64_t> level_off;  // n_levels+1 offsets into perm
    int64_t max_level = 0;           // largest level size
    int64_t n_levels() const { return (int64_t)level_off.size() - 1; }
};

// false if n does not fit the int32 permutation
bool build_level_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items,
                          int within_level_order, LevelSchedule &out);


// Hub-chain level schedule (see build_chain_schedule in level_schedule.cpp): a UNIT is a run of <= max_chain tuples that are
// consecutive in the CRS order of one hub row (an item, or a user) and whose other-side rows (spokes) are pairwise distinct;
// one 16-lane group walks a unit in order with the hub row resident in registers.  Units of a level are pairwise independent.
struct ChainSchedule {
    std::vector<int32_t> perm;      // stream position -> CRS tuple index; a unit's tuples are contiguous, in CRS order
    std::vector<int32_t> unit_off;  // n_units+1 offsets into perm
    std::vector<int64_t> level_off; // n_levels+1 offsets into unit_off (unit indices)
    int hub_is_item = 1;
    int64_t max_level_units = 0;    // most units in one level
    int64_t n_units() const { return (int64_t)unit_off.size() - 1; }
    int64_t n_levels() const { return (int64_t)level_off.size() - 1; }
};
// hub: 0 = chain along users (P[u] resident), 1 = along items (Q[j] resident), -1 = whichever gives fewer units, -2 / -3 = users / items
// unless that costs more than 1.3x the units of the other side (the side that carries the model's context-bias rows: kept on chip along a
// unit they cost one coalesced row per unit instead of scattered 4-byte read-modify-writes per tuple)
bool build_chain_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub, int max_chain,
                          ChainSchedule &out);

// Owner (dataflow) schedule for heavy-tailed degrees (see build_owner_schedule in level_schedule.cpp): every hub row (an item, or a
// user) belongs to ONE owner -- a wavefront of the persistent sgd_owner kernel -- which walks all tuples of its rows in CRS order with
// the current hub row in registers; the other side's rows (spokes) travel between owners through tagged records in HBM.
enum { OWN_HUB_FWD = 1, OWN_HUB_LATE = 2, OWN_HUB_STORE = 4, OWN_SPK_FWD = 8, OWN_SPK_STORE = 16,
       OWN_NOP = 32 /* list padding, added by cmi_set_ratings */ };
struct OwnerSchedule {
    std::vector<int32_t> perm;     // list position -> CRS tuple index; an owner's tuples are contiguous, in CRS order
    std::vector<int64_t> own_off;  // n_owners+1 offsets into perm
    std::vector<uint32_t> want;    // per position: how many earlier tuples of the epoch share its spoke row (= the tag it must carry)
    std::vector<uint32_t> flags;   // per position: OWN_* bits
    int hub_is_item = 1;
    int64_t max_load = 0;          // tuples of the busiest owner
    int64_t n_owners() const { return (int64_t)own_off.size() - 1; }
};
// hub: 0 = users are owned, 1 = items are owned, -1 = the side with the larger maximum degree.  depth = prefetch distance of the
// kernel (a hub row written fewer than depth+1 places back in the owner's list must be re-read late, OWN_HUB_LATE).
bool build_owner_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub, int n_owners,
                          int depth, OwnerSchedule &out);

// number of levels of the plain schedule (longest dependency chain), without building it
int64_t count_plain_levels(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items);

// Narrow runs of a level schedule: run_len[l] > 0 = a run of that many consecutive levels, each with <= max_tuples tuples,
// starts at level l (one launch walks it); -1 = inside such a run; 0 = the level is launched on its own.  Runs shorter than
// min_levels are not formed.  Returns the number of launches per epoch.
int64_t build_narrow_runs(const std::vector<int64_t> &level_off, int64_t max_tuples, int64_t min_levels, std::vector<int32_t> &run_len);

// Conflict-free CRS blocks (CAMF_C): maximal runs of consecutive tuples that share no user and no item, cut at max_block
// tuples.  off = block offsets (n_blocks + 1 entries).
void build_conflict_free_blocks(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int max_block,
                                std::vector<int32_t> &off);

} // namespace cmi
