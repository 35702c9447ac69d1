set -o pipefail
O=gpurun_out/pp6; mkdir -p $O
for pp in 0 2; do
  echo "== pp $pp" >> $O/out.txt
  LDM_GEMM_PINGPONG=$pp timeout -k 10 300 python -u tools/gemm_bench.py "geglu" "ff1 plain" --tile 256,128 --iters 30 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
cat $O/out.txt
