"""The real Frappe rating file (BASELINE.json configs[1]) for the tests: tests/golden/frappe_compact.csv.gz unpacked into a
temporary directory, as the raw file (rating = the usage count `cnt`, 1 .. 28 752) or with the rating column put on the
log scale the Frappe literature trains on (1 + log10(cnt), 6 decimals -- derived here, deterministically, from the committed raw file;
the raw counts make every SGD recommender of the reference overflow at its default learning rate, which is a test case of its own)."""
import gzip
import math
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SETTING = """dataset.ratings.lins=%(path)s
ratings.setup=-threshold -1 -datatransformation 1 -fullstat -1
recommender=camf_c
evaluation.setup=cv -k %(folds)d -p off --rand-seed 1 --test-view all
item.ranking=off -topN 10
output.setup=-folder CARSKit.Workspace -verbose off
num.factors=64
num.max.iter=15
learn.rate=%(lr)s -max -1 -bold-driver
reg.lambda=0.0001 -c 0.001
"""


def write_ratings(tmp_path, scale="raw"):
    os.makedirs(str(tmp_path), exist_ok=True)
    text = gzip.open(os.path.join(GOLDEN, "frappe_compact.csv.gz"), "rb").read().decode("utf-8")
    if scale == "log":
        lines = text.split("\n")
        out = [lines[0]]
        for ln in lines[1:]:
            if not ln:
                out.append(ln)
                continue
            f = ln.split(",")
            f[2] = "%.6f" % (1.0 + math.log10(int(f[2])))
            out.append(",".join(f))
        text = "\n".join(out)
    path = os.path.join(str(tmp_path), "frappe_%s.csv" % scale)
    open(path, "w", encoding="utf-8").write(text)
    return path


def write_conf(tmp_path, scale="raw", lr="2e-2", folds=5):
    path = write_ratings(tmp_path, scale)
    conf = os.path.join(str(tmp_path), "setting_%s_%d.conf" % (scale, folds))
    open(conf, "w").write(SETTING % {"path": path, "lr": lr, "folds": folds})
    return conf
