#!/bin/bash
# usage (GPU box): tools/soak_all.sh [n_cameras] [grid_res]  -- tools/soak.py over every octree flavour (build flags 0 DAG + embedded masks, 1 no DAG,
# 2 DAG with plain indices, 3 two-level bricks, 4 / 7 conservative voxelization), a fresh seed per flavour; stops at the first mismatch
cd ${GRAFT_REPO_ROOT:?run through gpurun}
n=${1:-3}; res=${2:-256}
for f in 0 1 2 3 4 7; do
  echo "== build flags $f"
  MVRT_SOAK_SEED=$((3000 + f)) timeout -k 10 400 python3 tools/soak.py $n $res $f || exit 1
done
