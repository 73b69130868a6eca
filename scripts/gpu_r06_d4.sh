#!/bin/bash
TAG=${1:-r06s}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dwdirect4.py tests/test_gpu_convc3rows.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8
python - <<'PY'
import torch, qnnpack_amd, bench
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
for shape in [(28, 28, 3, 3, 1, 1, 122, 1, 1), (56, 56, 3, 3, 2, 1, 122, 1, 1), (28, 28, 3, 3, 1, 1, 58, 1, 1), (56, 56, 3, 3, 2, 1, 58, 1, 1), (56, 56, 3, 3, 2, 1, 50, 1, 1)]:
    H, W, KH, KW, S, D, G, GIC, GOC = shape
    for v in (0, 1):
        lib.set_option("dwconv_kernel", v)
        layer = bench.ConvLayer(lib, torch, 128, H, W, KH, KW, S, D, G, GIC, GOC, seed=5, min_bytes_between_reuse=512 << 20)
        print(shape, layer.kernel, round(layer.time_ms(2, 10) * 1e3, 1), "us")
        layer.close()
    lib.set_option("dwconv_kernel", 0)
PY
