#!/bin/bash
# tools/probe/stamps/cw.so = the library with conv_wave.hip built -DCW_STAMPS (tools/probe/wave_stamps.py)
set -e
root="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$root"
python deepq-decoding_amd/build.py > /dev/null
mkdir -p tools/probe/stamps
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCW_STAMPS $CW_EXTRA -c deepq-decoding_amd/csrc/conv_wave.hip -o /tmp/cw_stamp.o
objs=""; for src in deepq-decoding_amd/csrc/*.hip; do o=$(basename $src .hip); if [ $o = conv_wave ]; then objs="$objs /tmp/cw_stamp.o"; else objs="$objs deepq-decoding_amd/lib/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/probe/stamps/cw.so $objs deepq-decoding_amd/lib/build_digest.o
