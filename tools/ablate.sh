set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q > gpurun_out/t4.log 2>&1 || true; tail -3 gpurun_out/t4.log
for ab in 0 1 15; do
  QM_ABLATE=$ab timeout -k 10 200 python - <<PY
import os,sys,time
sys.path.insert(0,'.')
import quasimodo_amd as q
eng=q.Engine(0); tid=eng.truth_synth(5_000_000,100_000,3)
nv=256
b=eng.batch([1_000_000]*nv,[tid]*nv); b.synth(5_000_000,100_000,3,3000)
for _ in range(3): b.run()
b.set_timing(True)
for _ in range(10): b.run()
t=b.timings()
byts=nv*(17e6+1.2e6)
print("ablate=%s classify %.3f ms (%.0f GB/s)  finalize %.3f  compact %.3f (%.0f GB/s out)"%(os.environ.get("QM_ABLATE"),t["classify_ms"],byts/t["classify_ms"]/1e6,t["finalize_ms"],t["compact_ms"],nv*3.7e6/t["compact_ms"]/1e6))
PY
done
