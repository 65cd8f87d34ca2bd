#!/bin/bash
# Build an experimental variant of libscint_hip.so without touching the tree:
#   tools/build_variant.sh <name> [patch | -DNAME=VALUE ...]      -> variants/<name>.so   (git-ignored, travels with gpurun)
# The patches (git diff format, relative to the repository root) are applied to a scratch copy of
# scintools_amd/ + include/; -D arguments are passed to every compile (then every unit is rebuilt).  On the GPU box an A/B swaps the library file:
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
flags=""; force=False
for p in "$@"; do
  case "$p" in
    -D*) flags="$flags $p"; force=True ;;
    *) (cd "$W" && patch -p1 -s < "$p") ;;
  esac
done
(cd "$W" && SCINT_VARIANT_FLAGS="$flags" python -c "from scintools_amd import build; build.build(force=$force, verbose=False)")
cp "$W/scintools_amd/libscint_hip.so" "$R/variants/$name.so"
echo "variants/$name.so"
