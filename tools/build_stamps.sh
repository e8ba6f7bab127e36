#!/bin/bash
# diagnostic build of the library with in-kernel stamps in sparse_conv_st.hip (loaded through PVX_LIB by tools/st_stamps.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include ${PVX_STAMP_DEF--DPVX_ST_STAMPS} $PVX_STAMP_FLAGS -c openpcdet_amd/csrc/sparse_conv_st.hip -o /tmp/abl/st_stamps.o
objs=$(ls openpcdet_amd/csrc/_obj/*.o | grep -v '/sparse_conv_st.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openpcdet_amd/lib/${PVX_STAMP_OUT-libpcdet_stamps.so} $objs /tmp/abl/st_stamps.o
