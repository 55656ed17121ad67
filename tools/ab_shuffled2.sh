#!/bin/bash
# like ab_shuffled.sh, with per-variant ENVIRONMENT too: "<tag>=<flags>[@VAR=value ...]"; two rounds interleaved
S=$PWD/quasimodo_amd/csrc
for spec in "$@"; do
  TAG=${spec%%=*}; REST=${spec#*=}; FLAGS=${REST%%@*}
  D=$PWD/gpurun_out/ab/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>/dev/null || echo "build failed: $TAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $S/qmvt_api.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for rep in 1 2; do for spec in "$@"; do
  TAG=${spec%%=*}; REST=${spec#*=}; ENVS=""; [[ "$REST" == *@* ]] && ENVS=${REST#*@}
  echo -n "$TAG: "; ( [ -n "$ENVS" ] && export $ENVS; QM_LIBQMVT=$PWD/gpurun_out/ab/$TAG/libqmvt.so python3 tools/join_ab.py 2>&1 | grep "ms per step" | sed -E 's/.*: ([0-9.]+ ms per step).*digest (.*)/\1 \2/' )
done; done
