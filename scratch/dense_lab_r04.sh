# A/B of dense-tracer store variants on the GPU box: rebuild with DRT_EXTRA_FLAGS, time the kernel (bench_dense.py)
# usage: VARIANTS="-DX|-DY" bash scratch/dense_lab_r04.sh
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VS <<< "${VARIANTS:-|-DDRT_STORE_LAB_PLAIN|-DDRT_DENSE_LAB_ONLYV|-DDRT_DENSE_LAB_OCC=6|-DDRT_DENSE_LAB_OCC=3}"
for v in "${VS[@]}"; do
  echo "=== variant: [$v]"
  touch differt_amd/csrc/trace_dense.hip
  DRT_EXTRA_FLAGS="$v" python -m differt_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python bench_dense.py --max-chunks 16 | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(r['kernel_ms_per_launch'], 'e2e', d['candidates_per_s'], 'fill', r['box_fill_GBps'])"
done
touch differt_amd/csrc/trace_dense.hip; python -m differt_amd.build > /dev/null 2>&1
