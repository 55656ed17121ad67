#!/bin/bash
# same-box A/B of kernel build variants on the SHUFFLED workload (bucket path): bash tools/ab_shuffled.sh "<tag>=<flags>" ...
S=$PWD/quasimodo_amd/csrc
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=$PWD/gpurun_out/ab/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>/dev/null || echo "build failed: $TAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $S/qmvt_api.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for spec in "$@"; do
  TAG=${spec%%=*}
  echo "== $TAG"; QM_LIBQMVT=$PWD/gpurun_out/ab/$TAG/libqmvt.so QM_SORT_PATH=buckets python3 - <<PY 2>&1 | grep -v amdgpu
import os, sys, time
sys.path.insert(0, ".")
import quasimodo_amd as q
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 3)
b = eng.batch([1_000_000] * 256, [tid] * 256)
b.synth(5_000_000, 100_000, 3, 3000, shuffled=True)
try:
    b.run(); b.finish()
    t0 = time.time()
    for _ in range(5):
        b.run(); b.finish()
    print("%.3f ms per step" % ((time.time() - t0) / 5 * 1e3))
except Exception as e:
    print("error", str(e)[:100])
PY
done
