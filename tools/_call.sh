set -u
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_tile" 2>&1 | tail -12 )
( GEMM_AB_KINDS=259,261 timeout 300 python tools/gemm_ab.py 7 ) 2>&1 | grep -v amdgpu
