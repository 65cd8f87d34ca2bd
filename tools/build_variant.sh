#!/bin/bash
# Build an experimental variant of libscint_hip.so without touching the tree:
#   tools/build_variant.sh <name> [patch ...]      -> variants/<name>.so   (git-ignored, travels with gpurun)
# The patches (git diff format, relative to the repository root) are applied to a scratch copy of
# scintools_amd/ + include/.  On the GPU box an A/B swaps the library file:
#   cp variants/<name>.so scintools_amd/libscint_hip.so
set -eu
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/vb/$name
rm -rf "$W"; mkdir -p "$W" "$R/variants"
cp -rp "$R/include" "$W/include"
mkdir -p "$W/scintools_amd"
cp -p "$R"/scintools_amd/*.py "$W/scintools_amd/"
cp -rp "$R/scintools_amd/csrc" "$W/scintools_amd/csrc"     # with the objects: only patched units are rebuilt
for p in "$@"; do (cd "$W" && patch -p1 -s < "$p"); done
(cd "$W" && python -c "from scintools_amd import build; build.build(verbose=False)")
cp "$W/scintools_amd/libscint_hip.so" "$R/variants/$name.so"
echo "variants/$name.so"
