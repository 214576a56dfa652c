#!/bin/bash
# A/B builds of the library beside the product one: scripts/build_variant.sh <name> [make variables ...]
#   scripts/build_variant.sh wino EXTRA=-DSK_SOME_SWITCH   -> build_alt/wino/sidekit_amd/csrc/libsidekit_amd.so
# The variant is a copy of csrc/ + include/ built in place (build_alt/ is git-ignored and travels to the GPU box); judge it with
# scripts/ab_pipelined.py on ONE box.
set -e
NAME=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
DST="$ROOT/build_alt/$NAME"
mkdir -p "$DST/sidekit_amd/csrc" "$DST/include"
cp "$ROOT"/include/*.h "$DST/include/"
cp "$ROOT"/sidekit_amd/csrc/*.hip "$ROOT"/sidekit_amd/csrc/*.cpp "$ROOT"/sidekit_amd/csrc/*.h "$ROOT"/sidekit_amd/csrc/Makefile "$DST/sidekit_amd/csrc/"
make -C "$DST/sidekit_amd/csrc" -j8 "$@" > "$DST/build.log" 2>&1 || { tail -30 "$DST/build.log"; exit 1; }
echo "$DST/sidekit_amd/csrc/libsidekit_amd.so"
