python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -5
python bench.py --steps 30 --warmup 5 2>&1 | tail -1
python - <<'PY'
import sys; sys.path.insert(0,'ground-plane-polling_amd')
import numpy as np, torch
from keras_retinanet_3D import models
m = models.load_model('synthetic:1234')
plan = m.plan_for(8, 402, 1333, 1000, True)
for k, v in plan.tuning.items(): print(k, v)
PY
