#!/bin/bash
# diagnostic build of the library with per-workgroup first / last ticks in the streaming and rolling conv kernels (tools/probes/fin_spread.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DPVX_FIN_STAMPS -c openpcdet_amd/csrc/sparse_conv.hip -o /tmp/abl/sc_fin.o
objs=$(ls openpcdet_amd/csrc/_obj/*.o | grep -v '/sparse_conv.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openpcdet_amd/lib/libpcdet_fin.so $objs /tmp/abl/sc_fin.o
