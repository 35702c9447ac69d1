set -o pipefail
mkdir -p gpurun_out/pp2
for abl in 0 1 8 4 2 12 6; do
  echo "== PP=1 ablate $abl" >> gpurun_out/pp2/stamps.txt
  LDM_GEMM_PINGPONG=1 LDM_GEMM_ABLATE=$abl timeout -k 10 100 python -u tools/gemm_stamps.py "L0 conv3x3 320" "L2 conv3x3" 2>&1 | grep "conv3x3" >> gpurun_out/pp2/stamps.txt || exit 1
done
cat gpurun_out/pp2/stamps.txt
