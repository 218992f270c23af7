#!/bin/bash
# usage (GPU box): tools/exp/host_at_scale.sh [million lines, multiple of 3]  -- the C++ host (carskit-mi355x) end to end on a generated
# binary-format rating file: read, 80/20 split, CAMF_CI k=64, 10 epochs, evalRatings.  Wall times of the whole run and of the reader.
set -e
cd "$(dirname "$0")/../.."
m=${1:-24}
python tools/exp/dao_read_time.py $m | tail -3
mkdir -p /tmp/host_scale/CARSKit.Workspace
ln -sf /tmp/dao_big.csv /tmp/host_scale/ratings.csv
ln -sf /tmp/dao_big.csv /tmp/host_scale/CARSKit.Workspace/train.csv      # -datatransformation -1: the file is in binary format already
cat > /tmp/host_scale.conf <<EOF
dataset.ratings.lins=/tmp/host_scale/ratings.csv
ratings.setup=-threshold -1 -datatransformation -1 -fullstat -1
recommender=camf_ci
evaluation.setup=given-ratio -r 0.8 --rand-seed 1 --test-view all
item.ranking=off -topN 10
output.setup=-folder CARSKit.Workspace -verbose on
num.factors=64
num.max.iter=10
learn.rate=2e-3 -max -1 -bold-driver
reg.lambda=0.0001 -c 0.001
EOF
( time carskit_amd/bin/carskit-mi355x -c /tmp/host_scale.conf ) 2>&1 | cut -c1-220 | tail -24
