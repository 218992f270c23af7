#!/bin/bash
# usage: tools/exp/sanitize_host.sh  -- builds data_dao.cpp + level_schedule.cpp (host-only translation units) with g++ under
# -fsanitize=thread and -fsanitize=address,undefined and runs them on a generated 400 K-line rating file (the ranged reader: 6 host
# threads), a file with a bad line (the fall-back to the sequential pass) and a 2 M-tuple hub-chain schedule.  No GPU.  Expect no report.
set -e
cd "$(dirname "$0")/../.."
out=/tmp/cmi_sanitize
mkdir -p $out
python - <<'PY'
import numpy as np
rng = np.random.default_rng(1)
n = 400_000
u, i, r = rng.integers(0, 200_000, n), rng.integers(0, 20_000, n), rng.integers(1, 6, n)
c = rng.integers(0, 4, (n, 2))
bits = np.array(["1,0,0,0", "0,1,0,0", "0,0,1,0", "0,0,0,1"])
with open("/tmp/cmi_sanitize/in.csv", "w") as f:
    f.write("User, Item, Rating, a:0, a:1, a:2, a:3, b:0, b:1, b:2, b:3\n")
    f.write("\n".join("u%d,i%d,%d,%s,%s" % t for t in zip(u.tolist(), i.tolist(), r.tolist(), bits[c[:, 0]].tolist(), bits[c[:, 1]].tolist())) + "\n")
open("/tmp/cmi_sanitize/bad.csv", "w").write("User,Item,Rating,a:x,a:y\r\nu1,i1,3,1,0\r\nu2,i1,abc,0,1\r\n")
PY
for san in thread address,undefined; do
  g++ -std=c++17 -O1 -g -fsanitize=$san -pthread tools/exp/sanitize_host_main.cpp carskit_amd/csrc/data_dao.cpp carskit_amd/csrc/level_schedule.cpp -o $out/host_$$
  echo "== -fsanitize=$san"
  CMI_HOST_THREADS=6 $out/host_$$ $out/in.csv
  CMI_DAO_PARALLEL_MIN_LINES=1 CMI_HOST_THREADS=4 $out/host_$$ $out/bad.csv | head -2
  rm -f $out/host_$$
done
