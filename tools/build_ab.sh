#!/bin/bash
# A/B variants of the library for one-box comparisons: tools/build_ab.sh <name> <file> "<flags>" [<file2> "<flags2>"]
#   -> tools/probe/ab/<name>.so = the regular build with <file>.hip recompiled under the extra flags.
# On the GPU box: DQ_LIB_PATH=tools/probe/ab/<name>.so python bench.py ...
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
name="$1"; shift
mkdir -p tools/probe/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
declare -A repl
while [ $# -ge 2 ]; do
  f="$1"; fl="$2"; shift 2
  /opt/rocm/bin/hipcc $FLAGS $fl -c deepq-decoding_amd/csrc/$f.hip -o /tmp/ab_${name}_$f.o
  repl[$f]=/tmp/ab_${name}_$f.o
done
objs=""
for src in deepq-decoding_amd/csrc/*.hip; do o=$(basename $src .hip); if [ -n "${repl[$o]}" ]; then objs="$objs ${repl[$o]}"; else objs="$objs deepq-decoding_amd/lib/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/probe/ab/$name.so $objs deepq-decoding_amd/lib/build_digest.o      # (+ the regular build's digest unit: dq_build_digest)
ls -la tools/probe/ab/$name.so
