#!/bin/bash
mkdir -p gpurun_out/r06
{
echo "== parity of the matrix-free strip kernel (OBVI_SCHUR_MF=1), all structure / parity tests except the host-vs-device slot table comparison"
OBVI_SCHUR_MF=1 timeout 1500 python -m pytest tests/test_gpu_structure.py tests/test_gpu_parity.py -q -m gpu -k "not slot_tables" 2>&1 | tail -15
} 2>&1 | tee gpurun_out/r06/mf_prototype_parity.txt
