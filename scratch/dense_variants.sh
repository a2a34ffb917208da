# A/B of dense-kernel variants on the GPU box: rebuild with DRT_EXTRA_FLAGS, run the driver-style bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dense_variants
for v in "" "-DDRT_LAB_AGPR_STORE" "-DDRT_LAB_SETPRIO" ${EXTRA_VARIANTS}; do
  touch differt_amd/csrc/ray_ops.hip
  DRT_EXTRA_FLAGS="$v" python -m differt_amd.build > /dev/null 2>&1 || echo BUILD FAILED "[$v]"
  for rep in 1 2; do
    python bench.py --steps 100 --warmup 10 --no-paths --no-scaling --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('variant [$v] kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4), 'literal_us', round(d['cfg2_literal']['us_per_launch_hipgraph'],2), 'batched', round(d['cfg2_batched']['hbm_frac'],3))"
  done
done
touch differt_amd/csrc/ray_ops.hip; python -m differt_amd.build > /dev/null 2>&1
python -m pytest tests/test_ray_ops_gpu.py -m gpu -q -x -k dense 2>&1 | tail -1
