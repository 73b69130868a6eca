#!/bin/bash
TAG=${1:-r06m}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_convc3rows.py "tests/test_gpu_reference_lists.py" -k "c3rows or 224x224_k7" -m gpu -q -p no:cacheprovider 2>&1 | tail -5
python - <<'PY'
import torch, qnnpack_amd, bench
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
for shape in [(224, 224, 3, 3, 2, 1, 1, 3, 24), (224, 224, 3, 3, 2, 1, 1, 3, 32)]:
    H, W, KH, KW, S, D, G, GIC, GOC = shape
    layer = bench.ConvLayer(lib, torch, 128, H, W, KH, KW, S, D, G, GIC, GOC, seed=5, min_bytes_between_reuse=512 << 20)
    print(shape, layer.kernel, round(layer.time_ms(2, 10) * 1e3, 1), "us")
    layer.close()
PY
