#!/bin/bash
# same-box A/B of kernel build variants: bash tools/ab_r2.sh "<tag>=<flags>" ...
# env: NV (VCFs, default 1000), ARGS (extra run_once.py arguments after n_vcf and runs, e.g. "0 30" = allele-extended batch)
NV=${NV:-1000}; ARGS=${ARGS:-}
S=$PWD/quasimodo_amd/csrc
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=$PWD/gpurun_out/ab/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>/dev/null || echo "build failed: $TAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $S/qmvt_api.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for r in 1 2 3; do
  for spec in "$@"; do
    TAG=${spec%%=*}
    echo -n "$TAG: "; QM_LIBQMVT=$PWD/gpurun_out/ab/$TAG/libqmvt.so python3 tools/run_once.py $NV 8 $ARGS 2>&1 | grep -v amdgpu.ids
  done
done
