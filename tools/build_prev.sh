#!/bin/bash
# build the library of the last commit into tools/bin/libdisyolo_prev.so (for tools/ab_bench.sh)
set -e
cd "$(dirname "$0")/.."
git stash -q
(cd dis-yolo_amd/csrc && make >/dev/null 2>&1)
cp dis-yolo_amd/libdisyolo_hip.so tools/bin/libdisyolo_prev.so
git stash pop -q
(cd dis-yolo_amd/csrc && touch *.hip && make 2>&1 | grep -E "error|warning" || true)
