#!/bin/bash
# Sends the tree to a GPU box and runs a command there; records the commit id first (the snapshot travels without .git,
# bench.py reads .bench_head).
#   tools/gpu.sh [timeout_s] '<command>'      e.g.  tools/gpu.sh 900 'python -m pytest tests -m gpu -x -q'
#   tools/gpu.sh [timeout_s]                  runs tools/_run.sh (a scratch script of the moment, git-ignored) if it exists
cd "$(dirname "$0")/.."
git rev-parse --short HEAD > .bench_head
T=600
if [[ "$1" =~ ^[0-9]+$ ]]; then T=$1; shift; fi
if [ $# -gt 0 ]; then CMD="$*"; elif [ -f tools/_run.sh ]; then CMD='bash tools/_run.sh'; else
  echo "usage: tools/gpu.sh [timeout_s] '<command>'" >&2; exit 64; fi
/usr/local/graft/bin/gpurun --timeout "$T" -- "$CMD"
