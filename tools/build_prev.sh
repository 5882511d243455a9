#!/bin/bash
# build the library of the previous commit into tools/bin/libdisyolo_prev.so (for tools/ab_bench.sh), in a
# throwaway worktree: the working tree (and any stash) is not touched
set -e
cd "$(dirname "$0")/.."
W=$(mktemp -d /tmp/disyolo_prev.XXXXXX)
git worktree add -q --detach "$W" "${1:-HEAD~1}"
trap 'git worktree remove --force "$W"' EXIT
make -C "$W/dis-yolo_amd/csrc" >/dev/null 2>&1
mkdir -p tools/bin
cp "$W/dis-yolo_amd/libdisyolo_hip.so" tools/bin/libdisyolo_prev.so
