python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from picles_amd import configs, _capi as K
from picles_amd.parallel import SlabModel
cfg=configs.box4096(n=4096)
m=SlabModel(cfg.model,0,1)
m.seed(); m.run_steps(cfg.Δt,5,K.STEP_ZERO_FIRST)
m.backend.enable_timing(2); m.run_steps(cfg.Δt,20,K.STEP_ZERO_FIRST); torch.cuda.synchronize()
t=m.backend.get_timing(); print('region', t['advance_ms']/t['advance_launches'], t['advance_launches'])
m.backend.enable_timing(1); m.run_steps(cfg.Δt,10,K.STEP_ZERO_FIRST); torch.cuda.synchronize()
t=m.backend.get_timing(); print(t); print(m.backend.get_timing_samples(0)); print(m.backend.get_timing_samples(1))
PY
