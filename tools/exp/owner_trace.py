"""GPU box, debug library built with `make TRACE=1 OUT=../lib/libcarskit_trace.so OBJDIR=../../build/csrc_trace`:
python tools/exp/owner_trace.py [k] [team]   -- where does an owner epoch that runs BESIDE other owner epochs first leave the lone run?

The instance under test (fp64, CAMF_CI, `cmi_set_device_share(3)`, teams as CMI_OWNER_TEAM / the automatic rule give them) trains two
epochs alone and two epochs beside two neighbours that run owner epochs continuously.  The SECOND epoch of each run is traced: every
owner writes, per list position, the rows it read (hub row, spoke row, hub context row, biases) and the rows it produced
(owner_kernels.hip, CMI_OWNER_TRACE).  The two traces have the same schedule, so they are compared position by position and every
difference is classified:
  compute   inputs bit-identical in both runs, outputs differ                      -> the owner's arithmetic / registers
  transfer  spoke input differs although the producing tuple's output is identical -> the record hand-off (tags, stores, loads)
  carry     hub input differs although the previous tuple of the list left an identical hub row -> registers between steps
  table     an input that comes from the model tables (first use in the epoch) differs -> the state between epochs
plus the inexact run's own consistency: input of a tuple == output of its producer (no reference to the lone run needed), and the
team loader's / storer's view of a row against the compute wave's."""
import os, sys, threading
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("CMI_LIB_PATH", os.path.join(ROOT, "carskit_amd", "lib", "libcarskit_trace.so"))
import numpy as np
from carskit_amd import capi, synth
from tests import util

OWNER, F64 = capi.FLAG_SCHED_OWNER, capi.FLAG_STATE_F64
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TEAM = None if len(sys.argv) < 3 or sys.argv[2] == "default" else sys.argv[2]
OUT = os.environ.get("TRACE_DIR", "/tmp")
ROWS = 12


def make(d, team):
    if team is None: os.environ.pop("CMI_OWNER_TEAM", None)
    else: os.environ["CMI_OWNER_TEAM"] = team
    st = synth.init_state("CAMF_CI", d, K, seed=5, dtype=np.float64)
    i = capi.Instance("CAMF_CI", K, d.n_users, d.n_items, d.n_conds, flags=OWNER | F64)
    i.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
    i.set_device_share(3)
    i.set_ratings(d.u, d.j, d.ctx, d.r, d.ctx_ptr, d.ctx_conds)
    i.set_states(st)
    return i


ds = [synth.generate(3000, 300, 3, 4, 120000, seed=500 + s, item_zipf=1.2) for s in (1, 2, 3)]


def run(beside, path):
    test = make(ds[0], TEAM)
    neigh = [make(ds[1], "0"), make(ds[2], "0")] if beside else []
    stop = threading.Event()
    def spin(i):
        while not stop.is_set(): i.train_epoch(util.LR)
    th = [threading.Thread(target=spin, args=(i,)) for i in neigh]
    for t in th: t.start()
    L = test.L
    test.train_epoch(util.LR)
    s1 = test.get_states()
    assert L.cmi_debug_owner_trace(test.h) == 0
    test.train_epoch(util.LR)
    assert L.cmi_debug_owner_trace_dump(test.h, path.encode()) == 0
    s2 = test.get_states()
    stop.set()
    for t in th: t.join()
    info = test.schedule_info()
    for i in [test] + neigh: i.close()
    return s1, s2, info


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def differs(A, B, rows):  # per position: any lane of the given rows differs bit for bit
    return np.any(bits(A[:, rows, :]) != bits(B[:, rows, :]), axis=(1, 2))


def main():
    pa, pb = os.path.join(OUT, "trace_alone.bin"), os.path.join(OUT, "trace_beside.bin")
    a1, a2, info = run(False, pa)
    print("instance under test:", info)
    for attempt in range(6):
        b1, b2, _ = run(True, pb)
        e1 = max(float(np.max(np.abs(a1[n] - b1[n]))) for n in a1)
        e2 = max(float(np.max(np.abs(a2[n] - b2[n]))) for n in a2)
        print("attempt %d: max |alone - beside| after epoch 1: %.3e, after epoch 2: %.3e" % (attempt, e1, e2))
        if e2 > 0: break
    else:
        print("no deviation in 6 attempts")
        return
    A = np.fromfile(pa).reshape(-1, ROWS, 64)
    B = np.fromfile(pb).reshape(-1, ROWS, 64)
    va, vb = A[:, 3, 5] == 1.0, B[:, 3, 5] == 1.0
    print("positions %d, traced alone %d, beside %d, same set: %s" % (len(A), va.sum(), vb.sum(), bool(np.all(va == vb))))
    meta_same = np.all(bits(A[:, 3, 1:5]) == bits(B[:, 3, 1:5]))
    print("list entries (off, hub, want, flags) identical in both traces:", bool(meta_same))
    v = va & vb
    off, hub, want, flags = (B[:, 3, c].astype(np.int64) for c in (1, 2, 3, 4))
    HUB_FWD, SPK_FWD = 1, 8
    team_pos = np.any(B[:, 8, :] != 0, axis=1) | np.any(B[:, 10, :] != 0, axis=1)   # positions a team loader / storer wrote
    print("positions handled by teams (loader/storer rows written):", int(team_pos.sum()))
    # an input / output = several rows: hub side rows 0, 2 and lane 7 of row 3; spoke side row 1 and lane 0 of row 3
    def side(T_, base):
        hubside = np.concatenate([T_[:, base + 0, :], T_[:, base + 2, :], T_[:, base + 3, 7:8]], axis=1)
        spoke = np.concatenate([T_[:, base + 1, :], T_[:, base + 3, 0:1]], axis=1)
        return bits(hubside), bits(spoke)
    Ahi, Axi = side(A, 0); Aho, Axo = side(A, 4)
    Bhi, Bxi = side(B, 0); Bho, Bxo = side(B, 4)
    d_hi, d_xi = np.any(Ahi != Bhi, axis=1) & v, np.any(Axi != Bxi, axis=1) & v
    d_ho, d_xo = np.any(Aho != Bho, axis=1) & v, np.any(Axo != Bxo, axis=1) & v
    print("positions whose hub input / spoke input / hub output / spoke output differ between the runs: %d / %d / %d / %d of %d"
          % (d_hi.sum(), d_xi.sum(), d_ho.sum(), d_xo.sum(), v.sum()))
    # producers
    pos = np.nonzero(v)[0]
    key = {(int(off[p]), int(want[p])): int(p) for p in pos}
    prod = np.full(len(B), -1, dtype=np.int64)
    for p in pos:
        if flags[p] & SPK_FWD: prod[p] = p - 1
        elif want[p] > 0: prod[p] = key.get((int(off[p]), int(want[p]) - 1), -1)
    has_prod = prod >= 0
    compute = v & ~d_hi & ~d_xi & (d_ho | d_xo)
    transfer = v & d_xi & has_prod & ~d_xo[np.maximum(prod, 0)]
    table_x = v & d_xi & ~has_prod
    carry = v & d_hi & ((flags & HUB_FWD) != 0)
    carry[1:] &= ~d_ho[:-1]
    carry[0] = False
    table_h = v & d_hi & ((flags & HUB_FWD) == 0)
    print("classified: compute %d (team %d), transfer %d (consumer team %d, producer team %d), carry %d (team %d), spoke-from-table %d, hub-from-table %d"
          % (compute.sum(), (compute & team_pos).sum(), transfer.sum(), (transfer & team_pos).sum(), team_pos[np.maximum(prod, 0)][transfer].sum(),
             carry.sum(), (carry & team_pos).sum(), table_x.sum(), table_h.sum()))
    # the inexact run on its own: is a tuple's spoke input what its producer put out?
    self_bad = v & has_prod & np.any(Bxi != Bxo[np.maximum(prod, 0)], axis=1)
    self_bad_a = v & has_prod & np.any(Axi != Axo[np.maximum(prod, 0)], axis=1)
    print("spoke input != producer's output within one run: beside %d (consumer team %d, producer team %d), alone %d"
          % (self_bad.sum(), (self_bad & team_pos).sum(), team_pos[np.maximum(prod, 0)][self_bad].sum(), self_bad_a.sum()))
    hub_self = v & ((flags & HUB_FWD) != 0)
    hub_self[1:] &= np.any(Bhi[1:] != Bho[:-1], axis=1)
    hub_self[0] = False
    print("hub input != previous tuple's hub output within the run beside: %d (team %d)" % (hub_self.sum(), (hub_self & team_pos).sum()))
    # team roles against the compute wave (run beside)
    nf = team_pos & ((flags & SPK_FWD) == 0)
    ld_bad = nf & (np.any(bits(B[:, 8, :]) != bits(B[:, 1, :]), axis=1) | (bits(B[:, 9, 0]) != bits(B[:, 3, 0])))
    st_bad = team_pos & (np.any(bits(B[:, 10, :]) != bits(B[:, 5, :]), axis=1) | (bits(B[:, 11, 0]) != bits(B[:, 7, 0])))
    print("team loader's row != compute wave's input: %d; team storer's row != compute wave's output: %d" % (ld_bad.sum(), st_bad.sum()))

    def show(name, mask, n=4):
        for p in np.nonzero(mask)[0][:n]:
            q = int(prod[p])
            print("  %s: position %d off %d hub %d want %d flags %d team %s | producer position %d (team %s)"
                  % (name, p, off[p], hub[p], want[p], flags[p], bool(team_pos[p]), q, bool(team_pos[q]) if q >= 0 else None))
            for label, X, Y in (("spoke in  A/B", A[p, 1], B[p, 1]), ("spoke out A/B", A[p, 5], B[p, 5]), ("hub in   A/B", A[p, 0], B[p, 0]),
                                ("hub out  A/B", A[p, 4], B[p, 4])):
                lanes = np.nonzero(bits(X) != bits(Y))[0]
                if len(lanes):
                    l = int(lanes[0])
                    print("    %s: %d lanes differ, first lane %d: %016x vs %016x" % (label, len(lanes), l, int(bits(X)[l]), int(bits(Y)[l])))
            if q >= 0:
                lanes = np.nonzero(bits(B[p, 1]) != bits(B[q, 5]))[0]
                if len(lanes):
                    l = int(lanes[0])
                    print("    beside: input vs producer's output: %d lanes differ, first lane %d: %016x vs %016x (alone: %016x)"
                          % (len(lanes), l, int(bits(B[p, 1])[l]), int(bits(B[q, 5])[l]), int(bits(A[p, 1])[l])))
                    # the store-data hazard's signature: the wrong words are the data of the producer's NEXT store (its bias granules)
                    got, put, bias = bits(B[p, 1])[lanes], bits(B[q, 5])[lanes], int(bits(B[q, 7])[0])
                    print("      lanes %s; high words equal: %s; wrong low words all == low word of the producer's bias %08x: %s; wrong high words == bias high word: %s"
                          % (lanes.tolist(), bool(np.all(got >> 32 == put >> 32)), bias & 0xffffffff,
                             bool(np.all((got & 0xffffffff) == (bias & 0xffffffff))), bool(np.all(got >> 32 == bias >> 32))))
    show("compute", compute); show("transfer", transfer); show("carry", carry); show("self-inconsistent", self_bad)
    show("spoke-from-table", table_x); show("hub-from-table", table_h)


main()
