#!/bin/bash
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
j() { "$@" --json 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/MSM = %.1f MSM/s' % (d['avg_instance_ms'], 1e3/d['avg_instance_ms']))"; }
for s in 0 1; do
  echo "== system runtime, MSM_AMD_STAGED_UPLOAD=$s"
  echo -n "batch pageable:        "; MSM_AMD_STAGED_UPLOAD=$s j $P 20 5 gpu 5 true --warmup 1
  echo -n "batch pageable+cache:  "; MSM_AMD_STAGED_UPLOAD=$s j $P 20 5 gpu 5 true --warmup 1 --bases-cache 1024
  echo -n "sequential h2c:        "; MSM_AMD_STAGED_UPLOAD=$s j $P 20 5 gpu 5 --warmup 1
  echo -n "sequential h2c+cache:  "; MSM_AMD_STAGED_UPLOAD=$s j $P 20 5 gpu 5 --warmup 1 --bases-cache 1024
done
python - <<'PY'
import importlib, sys, time, os
sys.path.insert(0, '.')
import torch   # binds the wheel's HIP 7.0 runtime first, like bench.py
m = importlib.import_module("metal-msm-gpu-acceleration_amd")
from oracle import c_oracle as co
n, inst = 1 << 20, 5
for staged in ("0", "1"):
    os.environ["MSM_AMD_STAGED_UPLOAD"] = staged
    cfg = m.setup_metal_state(0)
    hs, hp = [], []
    for j in range(inst):
        dp, ds = cfg.generate_instance(0xB2540000 + j, n, True)
        hp.append(cfg.to_host(dp, 64 * n)); hs.append(cfg.to_host(ds, 32 * n)); cfg.free(dp); cfg.free(ds)
    want = cfg.msm_batch(hs, hp, [n] * inst)
    for cache in (0, 64 * n * inst * 2):
        cfg.set_bases_cache(cache)
        cfg.msm_batch(hs, hp, [n] * inst)
        t0 = time.perf_counter()
        for _ in range(3):
            outs = cfg.msm_batch(hs, hp, [n] * inst)
        dt = time.perf_counter() - t0
        assert outs == want
        print(f"torch process (HIP 7.0), staged={staged} cache={'on' if cache else 'off'}: {inst * 3 / dt:.1f} MSM/s")
    cfg.close()
PY
