set -u
export TMPDIR=/tmp
( timeout 240 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_tile" 2>&1 | tail -3 )
( GEMM_AB_KINDS=192,193,259 timeout 300 python tools/gemm_ab.py 7 shards ) 2>&1 | grep -v amdgpu
