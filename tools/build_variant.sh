#!/bin/bash
# Build libdvs_<name>.so from a git revision (default HEAD) for same-call A/B runs with tools/variant_bench.py.
# usage: tools/build_variant.sh <name> [rev] [extra hipcc flags...]      (the library lands next to libdvs_hip.so; it is
# git-ignored, travels with the gpurun snapshot, and must be deleted before a round ends: the package never loads it)
set -e
name="$1"; rev="${2:-HEAD}"; shift; shift || true
root="$(cd "$(dirname "$0")/.." && pwd)"
wt="/tmp/dvs_variant_$name"
rm -rf "$wt"; git -C "$root" worktree prune
git -C "$root" worktree add -f --detach "$wt" "$rev" >/dev/null
make -C "$wt/dags_vae_search_amd/csrc" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value $*" >/dev/null
cp "$wt/dags_vae_search_amd/libdvs_hip.so" "$root/dags_vae_search_amd/libdvs_$name.so"
git -C "$root" worktree remove --force "$wt"
echo "built dags_vae_search_amd/libdvs_$name.so from $rev $*"
