#!/bin/bash
# Instrumented builds for tools/stamp_run.py: tools/probe/stamps/s<tag>.so = the library with -DDQ_STAMPS=<tag> in the file that
# carries that tag's stamps (1, 2, 11, 12, 20: fused.hip; 3, 4, 5, 6, 13, 14, 15, 23: fused_bwd.hip); the other objects are the regular build's.
# Usage: tools/build_stamps.sh 1 2 3 4      then on the GPU box: DQ_LIB_PATH=tools/probe/stamps/s4.so python tools/stamp_run.py 4
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
python deepq-decoding_amd/build.py > /dev/null
mkdir -p tools/probe/stamps
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for tag in "$@"; do
  case $tag in 21) f=fused_bwd ;; *) case $((tag % 10)) in 1|2|0) f=fused ;; *) f=fused_bwd ;; esac ;; esac
  extra=""; if [ $tag = 6 ]; then extra="-DDQ_STAMP_BLOCK=300"; fi      # tag 6: a RIDING environment workgroup of the dense backward's launch (blocks >= 256)
  ( /opt/rocm/bin/hipcc $FLAGS -DDQ_STAMPS=$tag $extra -c deepq-decoding_amd/csrc/$f.hip -o /tmp/stamp_${f}_$tag.o
    objs=""; for src in deepq-decoding_amd/csrc/*.hip; do o=$(basename $src .hip); if [ $o = $f ]; then objs="$objs /tmp/stamp_${f}_$tag.o"; else objs="$objs deepq-decoding_amd/lib/$o.o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/probe/stamps/s$tag.so $objs deepq-decoding_amd/lib/build_digest.o ) &
done
wait
ls -la tools/probe/stamps/
