#!/bin/bash
# Build the library of a git revision (default HEAD) into build_ab/head/ -- here, where .git lives --
# so that a GPU call can time it against the working tree on the SAME box:
#   QM_LIBQMVT=$PWD/build_ab/head/libqmvt.so python3 tools/run_once.py 256 8
set -e
REV=${1:-HEAD}
D=build_ab/head; mkdir -p $D build_ab/include
for f in qmvt_kernels.hip qmvt_api.cpp qmvt_host.cpp qmvt_dev.h; do git show $REV:quasimodo_amd/csrc/$f > $D/$f; done
git show $REV:include/qmvt.h > build_ab/include/qmvt.h
sed -i 's|../../include/qmvt.h|../include/qmvt.h|' $D/qmvt_api.cpp $D/qmvt_host.cpp
(cd $D && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -o k.o qmvt_kernels.hip 2>/dev/null \
  && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c -o a.o qmvt_api.cpp 2>/dev/null \
  && g++ -O3 -std=c++17 -fPIC -pthread -c -o h.o qmvt_host.cpp \
  && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libqmvt.so k.o a.o h.o -lz)
echo "built $D/libqmvt.so from $REV"
