#!/usr/bin/env python3
"""Experiment behind DESIGN.md section 7 ("c = 1 vs c = W ... to validate by RMSE"): which merge rule should the
user-sharded multi-GPU epoch use for the replicated item-side state,

    item_side = start + (sum over ranks of (local - start)) / c,      c = 1 ("sum") or c = W ("mean")?

CPU only: every rank is an in-process oracle (the order-exact C restatement of the reference loop) over its user shard; the
W = 1 run IS the reference's sequential algorithm.  Reports the held-out RMSE after E bold-driver epochs.

  python tests/exp_merge_rule.py [--epochs 30] [--k 32] [--model CAMF_CI]
Shapes:  strong = one data set split W ways by user;  weak = every rank brings its own users (per-rank data fixed, the
bench.py --gpus N shape), compared with the sequential run over the union."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carskit_amd import dist as cdist, synth  # noqa: E402
from tests import util  # noqa: E402

USER_SIDE = ("P", "userBias", "ucBias")


def run_sharded(model, train, test, k, world, rule, epochs, lr0=util.LR, seed=77, track=None, stop_at=None):
    """Train `epochs` bold-driver epochs with `world` in-process ranks; returns (test RMSE, losses, train RMSE).
    rule: "sum" | "mean" | "adaptive" | "adaptive_tr" | "weighted" (per item / per cell, every rank's move weighted by its share of
    that item's ratings: equals "mean" on uniform data, differs under skew) ; lr0 may be scaled by the caller (lr x sqrt(W), lr x W).
    track: a list that receives the TRAINING RMSE of the assembled model after every epoch; stop_at: stop as soon as it is <= this."""
    gm = float(train.r.sum() / np.count_nonzero(train.r))
    st = synth.init_state(model, train, k, seed=seed)
    ranks = []
    for r in range(world):
        shard, (lo, hi) = cdist.shard_by_user(train, r, world)
        s = {n: (a[lo:hi].copy() if n in USER_SIDE else a.copy()) for n, a in st.items()}
        ranks.append((util.c_oracle(model, shard, k, s, gm), lo, hi))
    names = cdist.ITEM_SIDE[model]
    c = 1.0 if rule == "sum" else float(world)
    n_j = np.bincount(train.j, minlength=train.n_items).astype(np.float64)
    conds = [train.ctx_conds[train.ctx_ptr[x]:train.ctx_ptr[x + 1]] for x in range(train.n_ctx)]
    n_jc = np.zeros((train.n_items, train.n_conds))
    if model not in util.TWO_D:
        for t in range(train.n):
            n_jc[train.j[t], conds[train.ctx[t]]] += 1
    # "weighted": rank r's share of the ratings of item j (of cell (j, cond))
    if rule == "weighted" and world > 1:
        w_j, w_jc = [], []
        for o, lo, hi in ranks:
            m = (train.u >= lo) & (train.u < hi)
            nj = np.bincount(train.j[m], minlength=train.n_items).astype(np.float64)
            w_j.append(nj / np.maximum(n_j, 1.0))
            njc = np.zeros((train.n_items, train.n_conds))
            if model not in util.TWO_D:
                for t in np.flatnonzero(m):
                    njc[train.j[t], conds[train.ctx[t]]] += 1
            w_jc.append(njc / np.maximum(n_jc, 1.0))

    def assembled():
        full = {n: ranks[0][0].state[n] for n in names}
        for n in st:
            if n in USER_SIDE:
                full[n] = np.concatenate([o.state[n].reshape(hi - lo, -1) for o, lo, hi in ranks]).reshape(st[n].shape)
        return full

    def train_rmse_now():
        trctx = None if model in util.TWO_D else train.ctx
        return util.c_oracle(model, train, k, assembled(), gm).eval_ratings(train.u, train.j, trctx, train.r, 1.0, 5.0)["RMSE"]

    lr, last = lr0, 0.0
    losses = []
    for it in range(1, epochs + 1):
        start = {n: ranks[0][0].state[n].copy() for n in names}
        loss = sum(o.epoch(lr) for o, _, _ in ranks)
        if world > 1:
            for n in names:
                delta = sum(o.state[n] - start[n] for o, _, _ in ranks)
                if rule in ("adaptive", "adaptive_tr"):
                    # scalar-quadratic model of one pass: a row that sees n ratings with curvature s per rating contracts its
                    # error by rho = exp(-lr*s*n); W stale passes of n/W ratings each then sum to (1-rho^(1/W))*W / (1-rho)
                    # times the sequential move
                    if n == "Q":
                        s2 = float(np.mean(np.concatenate([o.state["P"].ravel() for o, _, _ in ranks]) ** 2))
                        if rule == "adaptive_tr":
                            s2 *= k   # curvature along a rating's own direction p_u: |p_u|^2 (trace), the conservative estimate
                        a = lr * (s2 * n_j)[:, None]
                    elif n == "itemBias":
                        a = lr * n_j
                    else:
                        a = lr * n_jc
                    a = np.maximum(a, 1e-12)
                    cc = world * (-np.expm1(-a / world)) / (-np.expm1(-a))
                    merged = start[n] + delta / cc
                elif rule == "weighted":
                    merged = start[n].copy()
                    for (o, _, _), wj, wjc in zip(ranks, w_j, w_jc):
                        d = o.state[n] - start[n]
                        merged += d * (wj[:, None] if n == "Q" else wj if n == "itemBias" else wjc)
                else:
                    merged = start[n] + delta / c
                for o, _, _ in ranks:
                    o.state[n][...] = merged
        losses.append(loss)
        if not np.isfinite(loss):
            return float("nan"), losses, float("nan")
        if it > 1:   # IterativeRecommender.updateLRate, bold driver (IterativeRecommender.java:216-229)
            lr = lr * 1.05 if abs(last) > abs(loss) else lr * 0.5
        last = loss
        if track is not None or stop_at is not None:
            tr_now = train_rmse_now()
            if track is not None:
                track.append(tr_now)
            if stop_at is not None and tr_now <= stop_at:
                break
    # evaluate with the assembled global model
    full = assembled()
    ev = util.c_oracle(model, test, k, full, gm)
    tctx = None if model in util.TWO_D else test.ctx
    trctx = None if model in util.TWO_D else train.ctx
    tr = util.c_oracle(model, train, k, full, gm).eval_ratings(train.u, train.j, trctx, train.r, 1.0, 5.0)["RMSE"]
    return ev.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)["RMSE"], losses, tr


def time_to_rmse(args):
    """how many epochs does a W-rank run need to reach the TRAINING RMSE the sequential (W = 1) run has after
    `--epochs` epochs -- for the mean rule, the per-item weighted mean, and the mean with the local learning rate scaled by sqrt(W) / W
    (the bold driver keeps steering from there).  'near-linear in updates/s' is not near-linear in time-to-RMSE: this is the factor."""
    data = synth.generate(args.users, args.items, 4, 4, args.users * args.per_user, seed=5, item_zipf=args.item_zipf or None)
    train, test = synth.split(data, 0.2)
    base_track = []
    base_rmse, _, _ = run_sharded(args.model, train, test, args.k, 1, "sum", args.epochs, track=base_track)
    target = base_track[-1]
    print(json.dumps({"world": 1, "rule": "sequential", "epochs": args.epochs, "train_rmse": target, "test_rmse": base_rmse}), flush=True)
    out = []
    cap = 6 * args.epochs
    for world in (2, 4, 8):
        for name, rule, scale in (("mean", "mean", 1.0), ("weighted", "weighted", 1.0), ("mean, lr x sqrt(W)", "mean", world ** 0.5),
                                  ("mean, lr x W", "mean", float(world)), ("sum", "sum", 1.0)):
            tr = []
            rmse, losses, _ = run_sharded(args.model, train, test, args.k, world, rule, cap, lr0=util.LR * scale, track=tr, stop_at=target)
            reached = bool(tr and tr[-1] <= target)
            rec = {"world": world, "rule": name, "epochs_to_target": len(tr) if reached else None, "slowdown_vs_sequential":
                   (len(tr) / args.epochs) if reached else None, "train_rmse_at_stop": tr[-1] if tr else None, "test_rmse": rmse,
                   "loss_increases": int(np.sum(np.diff(losses) > 0)), "diverged": bool(not np.all(np.isfinite(losses)))}
            print(json.dumps(rec), flush=True)
            out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--model", default="CAMF_CI")
    ap.add_argument("--users", type=int, default=16000)
    ap.add_argument("--items", type=int, default=1600)
    ap.add_argument("--per-user", type=int, default=25)
    ap.add_argument("--rules", default="sum,mean,adaptive,adaptive_tr")
    ap.add_argument("--shapes", default="strong,weak")
    ap.add_argument("--time-to-rmse", action="store_true", help="epochs to reach the sequential run's training RMSE (table in DESIGN.md section 7)")
    ap.add_argument("--item-zipf", type=float, default=0.0)
    args = ap.parse_args()
    if args.time_to_rmse:
        return time_to_rmse(args)
    out = []
    for shape in args.shapes.split(","):
        for world in (1, 2, 4, 8):
            nu = args.users if shape == "strong" else args.users // 8 * world
            data = synth.generate(nu, args.items, 4, 4, nu * args.per_user, seed=5)
            train, test = synth.split(data, 0.2)
            for rule in (("sum",) if world == 1 else tuple(args.rules.split(","))):
                rmse, losses, train_rmse = run_sharded(args.model, train, test, args.k, world, rule, args.epochs)
                rec = {"shape": shape, "world": world, "rule": rule, "users": nu, "ratings": train.n,
                       "ratings_per_item_per_rank": train.n / args.items / world, "test_rmse": rmse, "train_rmse": train_rmse,
                       "final_loss": losses[-1], "loss_increases": int(np.sum(np.diff(losses) > 0))}
                print(json.dumps(rec), flush=True)
                out.append(rec)
    return out


if __name__ == "__main__":
    main()
