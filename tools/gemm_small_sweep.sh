for cfg in "" "4,2,1" "4,2,2" "4,4,1" "4,4,2" "8,2,1"; do
  echo "== VEON_GEMM_SMALL=$cfg"
  VEON_GEMM_SMALL=$cfg GEMM_CFGS=0 python tools/gemm_bench.py 2 2>&1 | grep -v amdgpu | sed 's/| auto.*torch/| torch/' | cut -c1-110
done
