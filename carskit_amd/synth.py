"""Synthetic (u, i, ctx, r) tuple generator of BASELINE.md section 3, shaped the way
DataDAO.readData (reference src/carskit/data/processor/DataDAO.java:198-354) would hand the tuples to
a recommender: inner ids in first-seen order, duplicates collapsed (last write wins, :342), tuples in
the CRS order of the (user-item pair x context) rating matrix that MatrixIterator yields.

Pure numpy, deterministic for a given seed (PCG64).  No reference code involved."""
from dataclasses import dataclass, field

import numpy as np

DEFAULT_SEED = 20260927


@dataclass
class RatingData:
    n_users: int
    n_items: int
    n_conds: int          # number of context conditions (columns >= 3 of the binary format)
    n_dims: int           # context dimensions = active conditions per rating
    u: np.ndarray         # int32 [n]  inner user id of each tuple, CRS order
    j: np.ndarray         # int32 [n]  inner item id
    ctx: np.ndarray       # int32 [n]  inner context-combination id
    r: np.ndarray         # float64 [n]
    ctx_ptr: np.ndarray   # int32 [n_ctx+1]
    ctx_conds: np.ndarray  # int32 [nnz]  ascending condition ids of each context combination
    min_rate: float = 1.0
    max_rate: float = 5.0
    meta: dict = field(default_factory=dict)
    empty_conds: object = None   # EmptyContextConditions (DataDAO.java:213-214): the ":na" condition of every dimension, header order

    @property
    def n(self):
        return int(self.r.shape[0])

    @property
    def n_ctx(self):
        return int(self.ctx_ptr.shape[0] - 1)

    def subset(self, idx):
        """Tuples idx (ascending positions keep CRS order); id spaces unchanged."""
        idx = np.asarray(idx)
        return RatingData(self.n_users, self.n_items, self.n_conds, self.n_dims, self.u[idx], self.j[idx],
                          self.ctx[idx], self.r[idx], self.ctx_ptr, self.ctx_conds, self.min_rate, self.max_rate,
                          dict(self.meta), self.empty_conds)


def _first_seen_ids(keys):
    """Map each key to the rank of its first occurrence (DataDAO: id = map size on first sight)."""
    uniq, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(uniq), dtype=np.int64)
    rank[order] = np.arange(len(uniq), dtype=np.int64)
    return rank[inv], len(uniq)


def generate(n_users, n_items, n_dims, conds_per_dim, n_ratings, seed=DEFAULT_SEED, item_zipf=None,
             latent_k=8, noise=0.5):
    """Generate ~n_ratings tuples (fewer after duplicate removal).

    u, i uniform (or items Zipf(item_zipf)); one active condition per dimension, uniform; integer
    ratings 1..5 from a latent_k model plus N(0, noise)."""
    rng = np.random.default_rng(seed)
    n = int(n_ratings)
    raw_u = rng.integers(0, n_users, n, dtype=np.int64)
    if item_zipf:
        w = 1.0 / np.arange(1, n_items + 1, dtype=np.float64) ** float(item_zipf)
        cdf = np.cumsum(w)
        cdf /= cdf[-1]
        raw_i = np.searchsorted(cdf, rng.random(n)).astype(np.int64)
        raw_i = rng.permutation(n_items)[raw_i]
    else:
        raw_i = rng.integers(0, n_items, n, dtype=np.int64)
    raw_c = rng.integers(0, conds_per_dim, (n_dims, n), dtype=np.int64) if n_dims else np.zeros((0, n), np.int64)
    ckey = np.zeros(n, dtype=np.int64)
    for d in range(n_dims):
        ckey = ckey * conds_per_dim + raw_c[d]
    n_ckeys = max(1, conds_per_dim ** n_dims)

    # ratings from a small latent model (chunked to bound memory)
    zu = rng.standard_normal((n_users, latent_k)).astype(np.float32)
    zi = rng.standard_normal((n_items, latent_k)).astype(np.float32)
    zc = rng.standard_normal(n_ckeys if n_ckeys < (1 << 24) else 1).astype(np.float32) * 0.3
    r = np.empty(n, dtype=np.float64)
    step = 1 << 22
    for s in range(0, n, step):
        e = min(n, s + step)
        dot = np.einsum("nk,nk->n", zu[raw_u[s:e]], zi[raw_i[s:e]]) / np.sqrt(latent_k)
        cb = zc[ckey[s:e]] if zc.shape[0] > 1 else 0.0
        val = 3.0 + dot + cb + noise * rng.standard_normal(e - s).astype(np.float32)
        r[s:e] = np.clip(np.rint(val), 1, 5)
    del zu, zi

    # duplicates of (u, i, ctx): the LAST line wins (dataTable.put overwrites, DataDAO.java:342)
    trip = (raw_u * n_items + raw_i) * n_ckeys + ckey
    rev = np.arange(n - 1, -1, -1)
    _, last_rev = np.unique(trip[rev], return_index=True)
    last_pos = rev[last_rev]                       # stream position of the surviving line per triple
    keep_val = r[last_pos]
    # the triple keeps the ids of its FIRST appearance; find the first position of each triple
    _, first_pos = np.unique(trip, return_index=True)  # same sorted-unique order as above
    order = np.argsort(first_pos, kind="stable")
    pos = first_pos[order]
    r = keep_val[order]
    raw_u, raw_i, ckey = raw_u[pos], raw_i[pos], ckey[pos]
    n = len(r)

    # first-seen inner ids (over the de-duplicated stream; every id's first sight is a first-appearance line)
    dense = lambda raw, size: _first_seen_dense(raw, size)[:2] if size <= (1 << 26) else _first_seen_ids(raw)   # same ids, no sort
    u, nu = dense(raw_u, n_users)
    i, ni = dense(raw_i, n_items)
    ui, n_ui = _first_seen_ids(raw_u * n_items + raw_i)
    ctx, n_ctx = dense(ckey, n_ckeys)

    # condition ids: column d*conds_per_dim + c; the context key lists them in ascending column order
    ctx_first = np.empty(n_ctx, dtype=np.int64)
    ctx_first[ctx[::-1]] = np.arange(n - 1, -1, -1)  # first stream position of each ctx id
    ck = ckey[ctx_first]
    conds = np.zeros((n_ctx, n_dims), dtype=np.int32)
    for d in range(n_dims - 1, -1, -1):
        conds[:, d] = d * conds_per_dim + (ck % conds_per_dim)
        ck = ck // conds_per_dim
    ctx_ptr = (np.arange(n_ctx + 1, dtype=np.int64) * n_dims).astype(np.int32)
    ctx_conds = conds.reshape(-1)

    # CRS order: user-item pair id ascending, then context id ascending
    order = np.lexsort((ctx, ui))
    return RatingData(int(nu), int(ni), int(n_dims * conds_per_dim), int(n_dims), u[order].astype(np.int32),
                      i[order].astype(np.int32), ctx[order].astype(np.int32), r[order], ctx_ptr, ctx_conds, 1.0, 5.0,
                      {"seed": seed, "n_ui": int(n_ui), "requested": int(n_ratings), "item_zipf": item_zipf})


def _mix64(x, key):
    """SplitMix64-style finaliser on uint64 arrays (wrap-around arithmetic)."""
    x = (x + np.uint64(key)) * np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def _distinct_indices(n, domain, seed):
    """n distinct pseudo-random integers in [0, domain): a 4-round Feistel permutation of 0..n-1 over the
    enclosing power-of-four domain, cycle-walked back into range (sampling without replacement, O(n))."""
    bits = max(2, int(domain - 1).bit_length())
    bits += bits & 1
    hb = np.uint64(bits // 2)
    mask = np.uint64((1 << (bits // 2)) - 1)
    keys = [int(k) for k in np.random.default_rng(seed).integers(1, 1 << 62, 4)]

    def perm(x):
        left, right = x >> hb, x & mask
        for k in keys:
            left, right = right, left ^ (_mix64(right, k) & mask)
        return (left << hb) | right

    def walk(lo, hi):        # every index's value is a pure function of the index: ranges are independent
        with np.errstate(over="ignore"):
            v = perm(np.arange(lo, hi, dtype=np.uint64))
            bad = np.flatnonzero(v >= np.uint64(domain))
            while len(bad):
                v[bad] = perm(v[bad])
                bad = bad[v[bad] >= np.uint64(domain)]
        out[lo:hi] = v.astype(np.int64)

    out = np.empty(n, dtype=np.int64)
    _ranges(n, 1 << 22, walk)
    return out


def _ranges(n, step, fn):
    """fn(lo, hi) over [0, n) in ranges of `step` on a few host threads (NumPy releases the GIL inside its loops).  Used only where
    the result does not depend on the split."""
    import concurrent.futures
    import os
    spans = [(lo, min(n, lo + step)) for lo in range(0, n, step)]
    workers = min(len(spans), max(1, min(16, (os.cpu_count() or 2) // 2)))
    if workers <= 1:
        for lo, hi in spans:
            fn(lo, hi)
        return
    done = set()

    def one(sp):
        fn(*sp)
        done.add(sp)

    try:
        with concurrent.futures.ThreadPoolExecutor(workers) as ex:
            list(ex.map(one, spans))
    except RuntimeError:            # "can't start new thread" (a container's pid limit): the remaining ranges on this thread
        for sp in spans:
            if sp not in done:
                fn(*sp)


def _first_seen_dense(raw, size):
    """First-seen inner ids for keys drawn from a small dense range [0, size): O(n) scatter instead of a sort."""
    n = len(raw)
    first = np.full(size, n, dtype=np.int64)
    first[raw[::-1]] = np.arange(n - 1, -1, -1, dtype=np.int64)  # repeated index: last write wins = first sight
    seen = np.flatnonzero(first < n)
    order = seen[np.argsort(first[seen], kind="stable")]
    rank = np.full(size, -1, dtype=np.int64)
    rank[order] = np.arange(len(order))
    return rank[raw], len(order), first[order]


def generate_fast(n_users, n_items, n_dims, conds_per_dim, n_ratings, seed=DEFAULT_SEED, latent_k=8, noise=0.5):
    """Large-scale variant of generate() for uniform users/items: the (user, item) pairs are sampled WITHOUT
    replacement (distributionally what "sample, then drop duplicates" gives), so no (u,i,ctx) duplicate can
    occur, every pair's id is its stream position and the CRS order IS the stream order -- no 50M-element sorts.
    Same output contract as generate()."""
    n = int(n_ratings)
    if n > n_users * n_items:
        raise ValueError("more ratings than (user, item) pairs")
    rng = np.random.default_rng(seed)
    pair = _distinct_indices(n, n_users * n_items, seed + 17)
    raw_u, raw_i = pair // n_items, pair % n_items
    del pair
    n_ckeys = max(1, conds_per_dim ** n_dims)
    ckey = rng.integers(0, n_ckeys, n, dtype=np.int64)

    zu = rng.standard_normal((n_users, latent_k)).astype(np.float32)
    zi = rng.standard_normal((n_items, latent_k)).astype(np.float32)
    zc = rng.standard_normal(n_ckeys).astype(np.float32) * 0.3
    r = np.empty(n, dtype=np.float64)
    step = 1 << 22
    inv = np.float32(1.0 / np.sqrt(latent_k))
    # the noise comes out of ONE generator stream, range after range (the values every earlier round generated); the gathers and the
    # arithmetic of a range are independent of the other ranges and run on a few host threads
    eps = np.empty(n, dtype=np.float32)
    for s in range(0, n, step):
        e = min(n, s + step)
        eps[s:e] = rng.standard_normal(e - s, dtype=np.float32)

    def rate(s, e):
        dot = np.einsum("nk,nk->n", zu[raw_u[s:e]], zi[raw_i[s:e]]) * inv
        val = 3.0 + dot + zc[ckey[s:e]] + noise * eps[s:e]
        r[s:e] = np.clip(np.rint(val), 1, 5)

    _ranges(n, step, rate)
    del zu, zi, eps

    u, nu, _ = _first_seen_dense(raw_u, n_users)
    i, ni, _ = _first_seen_dense(raw_i, n_items)
    ctx, n_ctx, ctx_first = _first_seen_dense(ckey, n_ckeys)
    ck = ckey[ctx_first]
    conds = np.zeros((n_ctx, n_dims), dtype=np.int32)
    for d in range(n_dims - 1, -1, -1):
        conds[:, d] = d * conds_per_dim + (ck % conds_per_dim)
        ck = ck // conds_per_dim
    ctx_ptr = (np.arange(n_ctx + 1, dtype=np.int64) * n_dims).astype(np.int32)
    return RatingData(int(nu), int(ni), int(n_dims * conds_per_dim), int(n_dims), u.astype(np.int32),
                      i.astype(np.int32), ctx.astype(np.int32), r, ctx_ptr, conds.reshape(-1), 1.0, 5.0,
                      {"seed": seed, "n_ui": n, "requested": n, "item_zipf": None, "generator": "fast"})


def split(data, test_ratio=0.2, seed=DEFAULT_SEED + 1):
    """Seeded 80/20 split (NOT the reference's RNG); both parts keep CRS order and the id spaces."""
    rng = np.random.default_rng(seed)
    mask = rng.random(data.n) < test_ratio
    idx = np.arange(data.n)
    return data.subset(idx[~mask]), data.subset(idx[mask])


def to_2d(data):
    """DataDAO.toTraditionalSparseMatrix (reference DataDAO.java:1241-1257): user x item matrix whose
    value is the mean over contexts of each (user, item) pair, iterated user ascending, item ascending.
    Returns (u, j, r) int32/int32/float64 in that CRS order."""
    key = data.u.astype(np.int64) * data.n_items + data.j
    uniq, inv = np.unique(key, return_inverse=True)
    sums = np.zeros(len(uniq))
    cnts = np.zeros(len(uniq))
    # sequential accumulation order within a pair = CRS order of the contextual matrix
    np.add.at(sums, inv, data.r)
    np.add.at(cnts, inv, 1.0)
    return (uniq // data.n_items).astype(np.int32), (uniq % data.n_items).astype(np.int32), sums / cnts


def java_float(x):
    """(double)(float)x : hyper-parameters are parsed as Java float (IterativeRecommender.java:36-40)."""
    return float(np.float32(x))


def init_state(model, data, k, seed=DEFAULT_SEED + 2, dtype=np.float64):
    """Initial parameter arrays in the reference's shapes and distributions (P,Q ~ N(0,0.1) then the
    model's bias containers in source order; icBias/ucBias of CAMF_CI/CU ~ U(0,1)); the stream is numpy's,
    not librec's unseeded java.util.Random -- parity runs inject these arrays on both sides."""
    rng = np.random.default_rng(seed)
    g = lambda *s: (0.1 * rng.standard_normal(s)).astype(dtype)
    st = {"P": g(data.n_users, k), "Q": g(data.n_items, k)}
    if model in ("BiasedMF", "CAMF_C"):
        st["userBias"], st["itemBias"] = g(data.n_users), g(data.n_items)
        if model == "CAMF_C":
            st["condBias"] = g(data.n_conds)
    elif model == "CAMF_CI":
        st["userBias"] = g(data.n_users)
        st["icBias"] = rng.random((data.n_items, data.n_conds)).astype(dtype)
    elif model == "CAMF_CU":
        st["itemBias"] = g(data.n_items)
        st["ucBias"] = rng.random((data.n_users, data.n_conds)).astype(dtype)
    elif model == "CAMF_CUCI":
        st["ucBias"] = g(data.n_users, data.n_conds)
        st["icBias"] = g(data.n_items, data.n_conds)
    elif model == "PMF":
        pass
    elif model == "SVD++":                       # SVDPlusPlus.java:46-53: BiasedMF.initModel, then Y ~ N(0, 0.1)
        st["userBias"], st["itemBias"] = g(data.n_users), g(data.n_items)
        st["Y"] = g(data.n_items, k)
    elif model == "CAMF_ICS":                    # CAMF_ICS.java:36-51: P, Q re-drawn uniform(0,1); all similarities start at 1
        st["P"], st["Q"] = rng.random((data.n_users, k)).astype(dtype), rng.random((data.n_items, k)).astype(dtype)
        st["ccMatrix"] = np.ones((data.n_conds, data.n_conds), dtype=dtype)
    elif model == "CAMF_LCS":                    # CAMF_LCS.java:34-41: cfMatrix_LCS ~ uniform(0,1), numF columns
        st["cfMatrix"] = rng.random((data.n_conds, int(data.meta.get("num_f", 10)))).astype(dtype)
    elif model == "CAMF_MCS":                    # CAMF_MCS.java:41-52: positions ~ uniform(0, 1/sqrt(numContextDims))
        st["cVector"] = (rng.random(data.n_conds) / np.sqrt(max(1, data.n_dims))).astype(dtype)
    else:
        raise ValueError(model)
    return st


def merge_user_parts(parts):
    """One data set out of per-shard parts that each own their users: part r's users become [sum of the earlier parts' user counts, ...),
    items keep their ids (the shared, replicated side), and the parts' CONTEXT ids -- each part numbers its context combinations in its
    own first-seen order -- are re-mapped onto ONE table: combinations are identified by their condition lists and numbered in
    first-seen order over the concatenated stream (what DataDAO would do with the concatenated file, DataDAO.java:281-290,330-333).
    Tuples stay in part order, so the result is in CRS order."""
    if not parts:
        raise ValueError("no parts")
    n_conds, n_dims = parts[0].n_conds, parts[0].n_dims
    table, rows = {}, []
    ctx_out, u_out, base = [], [], 0
    for p in parts:
        if p.n_conds != n_conds:
            raise ValueError("parts disagree on the number of conditions")
        ptr, conds = np.asarray(p.ctx_ptr, dtype=np.int64), np.asarray(p.ctx_conds, dtype=np.int32)
        remap = np.empty(p.n_ctx, dtype=np.int32)
        for c in range(p.n_ctx):               # the part's own ids ascend in ITS first-seen order: walking them in id order keeps that order
            key = conds[ptr[c]:ptr[c + 1]].tobytes()
            g = table.get(key)
            if g is None:
                g = table[key] = len(rows)
                rows.append(conds[ptr[c]:ptr[c + 1]])
            remap[c] = g
        ctx_out.append(remap[p.ctx])
        u_out.append(p.u.astype(np.int64) + base)
        base += p.n_users
    ctx_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    ctx_ptr[1:] = np.cumsum([len(x) for x in rows])
    cat = lambda name: np.concatenate([getattr(p, name) for p in parts])
    return RatingData(int(base), max(p.n_items for p in parts), n_conds, n_dims, np.concatenate(u_out).astype(np.int32), cat("j"),
                      np.concatenate(ctx_out).astype(np.int32), cat("r"), ctx_ptr.astype(np.int32),
                      np.concatenate(rows).astype(np.int32) if rows else np.zeros(0, np.int32), parts[0].min_rate, parts[0].max_rate,
                      {"merged_parts": len(parts), "part_users": [p.n_users for p in parts]}, parts[0].empty_conds)
