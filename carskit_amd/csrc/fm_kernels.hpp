// fm_kernels.hpp -- launch interface of fm_kernels.hip (internal)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmi {

// One rating as a phase streams it: the error as computed by cmi_fm_init (never rewritten by a sweep) and the rating's
// OTHER two feature ids.  a = item (user order) / user (item order) / user (context order); c = context-combination id
// (user and item order) / item (context order).
struct FmRec {
    double err0;
    int32_t a, c;
};

// Field 2 (context features; a few ratings in a thousand have one): the ratings of the field's support sorted by feature, 16-byte records;
// one wave per feature adds its piece (fm_ctx_kernel).  piece_off[l] .. piece_off[l + 1] = the records of feature l.
struct FmOrder {
    const FmRec *rec;
    const int32_t *piece_off; // count + 1
    int32_t count;
    int64_t n_rec;
};

// ---- fields 0 (users) and 1 (items): the CELL stream (round 5) -------------------------------------------------------------------
// What bounded the reduce launch of round 4 was the L2 REQUEST rate of the table gathers: one 16-byte gather per record, every lane
// of a wave in a line of its own (tools/micro/gather16.hip: 210 G such gathers/s on this part, 119 us for 25 M; the TCP sends ONE
// request for the lanes of an instruction that fall into the same 128-byte line: two lanes per line 57 us, four 36 us).  So the
// records a workgroup evaluates together are sorted BY THE GATHERED ID: a GROUP of coordinates (as many as a workgroup's register
// accumulators hold: ~4 900 users at C4's share = 3 lanes per line) is walked by H workgroups, each taking alternate sub-slices of the
// other field (fm_api.cpp fm_build_cells); the records of (group part, sub-slice) -- a CELL -- are cut into BATCHES of <= FMC_RCAP records
// in gathered-id order.  The workgroup evaluates a batch in that order (coalesced 8 + 4 byte streams, one gather per record), parks
// {e', h} in LDS at the record's position in COORDINATE order (`pos`, 14 bits of the packed word), and a thread per coordinate slot adds
// its run left to right into accumulators that live in its REGISTERS for the whole block: no per-piece partial sums go through memory,
// and a coordinate with one slot is updated by the same launch.  A run longer than FMC_RUN records inside a batch (hot coordinate) is
// spread over several slots; a coordinate with several slots -- H > 1, a hot run, records spanning blocks -- is COMPLEX: its slots' sums
// go to `partial3` and fm_cplx_kernel finishes it.
constexpr int FMC_THREADS = 1024;
constexpr int FMC_RCAP = 8192;  // records per batch: 128 KB of LDS parking (10 240 = all of a CU's LDS measured the same)
constexpr int FMC_SLOTS = 5120; // accumulator slots per block: kept in the threads' registers (3 doubles per slot)
constexpr int FMC_RUN = 64;     // longest run one thread adds
constexpr int FMC_CHUNK = 256;  // atomic form: a batch is a chunk of <= 256 records, the unit a WAVE takes (no barriers between chunks)

struct FmBatch {
    int32_t rec0, n; // records [rec0, rec0 + n) of the stream, in gathered-id order
    int32_t tab0;    // table entry (absolute index into FmArgs::tab) of the first id of the batch's id range; a batch of ratings WITH a
                     // context feature (n_flag = n): first entry of the batch in the side arrays fo / fcx instead
    int32_t poff0;   // poff[poff0 + s] .. poff[poff0 + s + 1]: parked positions of slot s (s local to the block)
    int32_t n_flag, pad;
};

struct FmCells {
    double *err0;           // n_rec: the error as computed by cmi_fm_init, stream order
    const uint32_t *pk;     // n_rec: bits 0..16 gathered id - first id of the batch's range, bits 17..30 parked position (deterministic form) or
                            // the record's slot inside its block (atomic form)
    const int32_t *fo, *fcx; // ratings with a context feature only (compact, a block's are contiguous): gathered id, context-combination id
    const FmBatch *bat;
    const int32_t *bat_off;    // n_blocks + 1: a block's batches, the pipelined ones first
    const int32_t *flag0;      // n_blocks: a block's first batch of ratings with a context feature (= bat_off[b + 1] if it has none)
    const uint16_t *poff;
    const int32_t *slot_off;   // n_blocks + 1
    const int32_t *slot_coord; // n_slots: coordinate, bit 31 = complex
    double *partial3;          // 3 x n_slots (complex slots only are written)
    double *w0part;            // n_slots: the w0 phase's per-slot sums
    const int32_t *cplx;       // 4 x n_cplx: coordinate, first slot, parts | slots per part << 16, stride between parts
    int32_t n_blocks, n_slots, n_cplx, count, S;
    int64_t n_rec;
};

struct FmArgs {
    // model (fp64, the reference's precision)
    double *w0;   // 1
    double *d0;   // 1: sum of the w0 deltas since cmi_fm_init
    double *w;    // p
    double *V;    // p x k row-major (the API's layout; cmi_fm_init, predict and the rankings read it)
    double *Vt;   // k x p: the sweeps' working copy -- a column of V is contiguous here (fm_col_load, fm_finish_kernel)
    // per-coordinate working table: .x = column f of V (the factor being swept), .y = D[l] = sum of the coordinate's
    // deltas (over w and every column of V) since cmi_fm_init.  The reference's errors[i] (FM.java:133-136,165,188,208) is
    //   err0[i] + d0 + D[user] + D[item] + xc * D[ctx feature]
    // so no phase ever writes per-rating data.
    double2 *tab;
    FmOrder ord[3]; // field 2 (context features): the per-wave chunk stream; fields 0 / 1: only `count` is set (the cell stream below)
    FmCells cell[2]; // field 0 users, 1 items
    double *part;   // [num | den] of the phase's field, or w0 scratch
    // the ratings in the caller's order as plain arrays (init only); E = err0 per rating, spread into the three streams
    const int32_t *u, *j, *ctx;
    const double *r;
    double *E;
    const int32_t *src[3]; // stream position -> rating, per field (init only)
    int64_t n, global_size;
    int32_t k, n_users, n_items, n_conds;
    int32_t atomic; // 0 (default): the packed words hold POSITIONS and fm_cell_kernel runs (fixed order); 1 (CMI_FM_FLAG_RELAXED_SUMS): slots, fm_cell_atomic_kernel
    int32_t xcol; // the column of V an UPDATE leaves in tab[].x of the coordinates it updates (-1: .x stays): see fm_update
    double xc; // 1 / numContextDims
    double regLw, regLf;
};

// f < 0: linear weights w; f >= 0: column f of V (a.tab[].x must hold it).
// mode 0: reduce -> part ([num | den] per coordinate: the exchange point of a multi-GPU host); 2: reduce + update (no exchange point)
hipError_t fm_launch_phase(const FmArgs &a, int field, int f, int mode, hipStream_t s);
hipError_t fm_launch_apply(const FmArgs &a, int field, int f, hipStream_t s);               // part -> update
hipError_t fm_launch_reduce_only(const FmArgs &a, int field, int f, hipStream_t s);         // the dominant kernel alone (timing; writes scratch only)
hipError_t fm_launch_col_load(const FmArgs &a, int field, int f, hipStream_t s);            // tab[l].x = Vt[f][l] for the coordinates of one field
hipError_t fm_launch_transpose(const double *src, double *dst, int64_t rows, int64_t cols, hipStream_t s); // dst[c][r] = src[r][c]
hipError_t fm_launch_w0_reduce(const FmArgs &a, double *scratch, hipStream_t s);            // part[0] = sum(err_i - w0)
hipError_t fm_launch_w0_apply(const FmArgs &a, hipStream_t s);
hipError_t fm_launch_init(const FmArgs &a, hipStream_t s);
hipError_t fm_launch_predict(const FmArgs &a, int64_t n, const int32_t *tu, const int32_t *tj, const int32_t *tc,
                             int bound, double lo, double hi, double *out, hipStream_t s);

} // namespace cmi
