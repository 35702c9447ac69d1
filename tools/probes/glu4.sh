set -o pipefail
O=gpurun_out/glu4; mkdir -p $O
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export LDM_GEMM_GLU_BK32=1; else unset LDM_GEMM_GLU_BK32; fi
  echo "== GLU_BK32=$v" >> $O/out.txt
  timeout -k 10 300 python -u tools/gemm_bench.py "L0 conv3x3 320" "geglu" --iters 40 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
cat $O/out.txt
