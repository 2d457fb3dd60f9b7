#!/bin/bash
# Builds libeodiff.so of another commit of THIS repository into scratch/altlib/libeodiff_<name>.so (git-ignored, but part of the
# snapshot gpurun sends), for same-box A/B runs against the working tree:   bash tools/build_lib_at.sh 5a34df7 r3
# Run in the build container (the GPU box has no .git); then:  gpurun -- "bash tools/same_box_ab.sh r3"
set -e
C=${1:?commit}; NAME=${2:?name}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$C" eo_diffusion_amd/csrc include | tar -x -C "$TMP"
make -j4 -C "$TMP/eo_diffusion_amd/csrc" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/scratch/altlib"
cp "$TMP/eo_diffusion_amd/lib/libeodiff.so" "$ROOT/scratch/altlib/libeodiff_$NAME.so"
rm -rf "$TMP"
echo "scratch/altlib/libeodiff_$NAME.so = commit $C"
