#!/bin/bash
# tools/probe/stamps/c16.so = the library with conv_bwd16.hip built -DC16_STAMPS (tools/probe/c16_stamps.py)
set -e
root="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$root"
python deepq-decoding_amd/build.py > /dev/null
mkdir -p tools/probe/stamps
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DC16_STAMPS $C16_EXTRA -c deepq-decoding_amd/csrc/conv_bwd16.hip -o /tmp/c16_stamp.o
objs=""; for src in deepq-decoding_amd/csrc/*.hip; do o=$(basename $src .hip); if [ $o = conv_bwd16 ]; then objs="$objs /tmp/c16_stamp.o"; else objs="$objs deepq-decoding_amd/lib/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/probe/stamps/c16.so $objs deepq-decoding_amd/lib/build_digest.o
