set -o pipefail
O=gpurun_out/pp13; mkdir -p $O
for lib in prev new prev new; do
  if [ $lib = prev ]; then export LDM_AMD_LIB=latent-diffusion_amd/build/prev/libldm_amd_prev.so; else unset LDM_AMD_LIB; fi
  echo "== $lib" >> $O/out.txt
  timeout -k 10 300 python -u tools/gemm_bench.py "conv3x3" "L0 ffout" "L0 qkv" --iters 30 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
paste -d'|' <(grep -A9 "== prev" $O/out.txt | head -10 | cut -c1-52,70-100) <(grep -A9 "== new" $O/out.txt | head -10 | cut -c70-100) <(grep -A9 "== prev" $O/out.txt | tail -10 | cut -c70-100) <(grep -A9 "== new" $O/out.txt | tail -10 | cut -c70-100)
for lib in prev new prev new; do
  if [ $lib = prev ]; then export LDM_AMD_LIB=latent-diffusion_amd/build/prev/libldm_amd_prev.so; else unset LDM_AMD_LIB; fi
  LDM_BENCH_AB=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_$lib.log 2>&1 || { tail $O/bench_$lib.log; exit 1; }
  echo "$lib $(grep -h '"metric"' $O/bench_$lib.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["achieved"])')"
done
